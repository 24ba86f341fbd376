"""The stride-1 3x3 convolutions of the R-101 bf16 backbone (batch 16, 800 x 1344) on the own MFMA
kernel against the library convolution + its epilogue pass; for the narrow layers (Cout <= 128) the
64-pixel-tile variant (1, 2, 2) against the 128-pixel one (2, 2, 2, IA_CONV3_VARIANT=22)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
import torch, torch.nn.functional as F
from iouaware import ops
def bench(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 16
for C, H, W in ((64, 200, 336), (128, 100, 168), (256, 50, 84), (512, 25, 42)):
    x = torch.randn(B, C, H, W, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device='cuda') * 0.03).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, device='cuda')
    wp = ops.conv3x3_bf16_pack(w)
    def lib():
        return ops.channel_affine_act_(F.conv2d(x, w, None, 1, 1), None, b, relu=True)
    def mine():
        return ops.conv3x3_bf16(x, wp, b, C, relu=True)
    fl = 2.0 * B * H * W * C * C * 9
    t0, t1 = bench(lib), bench(mine)
    os.environ['IA_CONV3_VARIANT'] = '22'
    t2 = bench(mine)
    del os.environ['IA_CONV3_VARIANT']
    print('%3d -> %3d  %3dx%3d  library %.3f ms (%.0f TF)   own %.3f ms (%.0f TF)   own, variant 22 forced %.3f ms' % (
        C, C, H, W, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t2), flush=True)
