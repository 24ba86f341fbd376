import sys, os
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
import torch
from iouaware import ops
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for shape in [(8, 256, 200, 336), (8, 64, 400, 672), (8, 256, 100, 168), (8, 256, 25, 42), (8, 2048, 25, 42), (8, 256, 7, 11)]:
    x = torch.randn(shape, device='cuda'); r = torch.randn(shape, device='cuda')
    C = shape[1]; s = torch.randn(C, device='cuda'); b = torch.randn(C, device='cuda')
    xc = x.contiguous(memory_format=torch.channels_last); rc = r.contiguous(memory_format=torch.channels_last)
    gb = x.numel() * 4 / 1e9
    a = t(lambda: ops.channel_affine_act_(x, s, b, relu=True)); a2 = t(lambda: ops.channel_affine_act_(x, s, b, residual=r, relu=True))
    c = t(lambda: ops.channel_affine_act_(xc, s, b, relu=True)); c2 = t(lambda: ops.channel_affine_act_(xc, s, b, residual=rc, relu=True))
    print('%-22s %.1f MB  NCHW %.1f us (%.2f TB/s) +res %.1f us (%.2f) | NHWC %.1f us (%.2f TB/s) +res %.1f us (%.2f)' % (
        shape, gb * 1e3, a, 2 * gb / a * 1e3, a2, 3 * gb / a2 * 1e3, c, 2 * gb / c * 1e3, c2, 3 * gb / c2 * 1e3))
