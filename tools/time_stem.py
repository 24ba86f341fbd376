"""The stem convolution (7x7 / 2, 3 -> 64, batch 8, 800 x 1344 fp32 channels-last): the own MFMA
kernel (csrc/stem.hip) against the library convolution in immediate mode."""
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
for B in (8, 1):
    x = torch.randn(B, 3, 800, 1344, device='cuda').contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 7, 7, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last)
    wp = ops.stem_weight(w)
    t0 = bench(lambda: F.conv2d(x, w, None, 2, 3))
    t1 = bench(lambda: ops.stem_conv(x, wp))
    fl = 2.0 * B * 400 * 672 * 64 * 147
    err = float((ops.stem_conv(x, wp) - F.conv2d(x, w, None, 2, 3)).abs().max())
    print('B=%d  library %.3f ms (%.0f TF)   own %.3f ms (%.0f TF, %.2f TB/s written)   max diff %.2e'
          % (B, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, B * 400 * 672 * 64 * 4 / t1 / 1e9, err), flush=True)

# bf16 (config 3): batch 16
x = torch.randn(16, 3, 800, 1344, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 3, 7, 7, device='cuda') * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
wp = ops.stem_weight_bf16(w)
t0 = bench(lambda: F.conv2d(x, w, None, 2, 3))
t1 = bench(lambda: ops.stem_conv_bf16(x, wp))
print('bf16 B=16  library %.3f ms   own %.3f ms (%.2f TB/s written)' % (t0, t1, 16 * 400 * 672 * 64 * 2 / t1 / 1e9), flush=True)
