"""bf16 3x3 tower convolution (256 -> 256, batch 16, the five pyramid levels of 800x1344): the
library's own MFMA implicit-GEMM kernel (csrc/conv3x3_bf16.hip, bias + ReLU fused) against
MIOpen / CK's convolution + the separate bias + ReLU pass."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
import torch, torch.nn.functional as F
from iouaware import ops
torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
def bench(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = [0.0, 0.0]
for (H, W) in [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]:
    x = torch.randn(B, 256, H, W, device='cuda')
    kind = os.environ.get('IA_BENCH_INPUT', 'randn')        # randn | relu (what a tower layer really reads) | zeros
    x = x.clamp(min=0) if kind == 'relu' else (x * 0 if kind == 'zeros' else x)
    x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, 256, 3, 3, device='cuda') * 0.03).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(256, device='cuda')
    wp = ops.conv3x3_bf16_pack(w)
    def lib():
        y = F.conv2d(x, w, None, 1, 1)
        return ops.channel_affine_act_(y, None, b, relu=True)
    def mine():
        return ops.conv3x3_bf16(x, wp, b, 256, relu=True)
    t0, t1 = bench(lib), bench(mine)
    fl = 2.0 * B * H * W * 256 * 256 * 9
    tot[0] += t0; tot[1] += t1
    var = ''
    for v in ('41', '21'):                      # forced variants (4, 1, 4) / (2, 1, 4): 128- / 64-pixel tiles
        os.environ['IA_CONV3_VARIANT'] = v
        tv = bench(mine)
        var += '  %s: %.3f (%.0f)' % (v, tv, fl / tv / 1e9)
    del os.environ['IA_CONV3_VARIANT']
    print('%3dx%3d  library conv + epilogue %.3f ms (%.0f TF)   own kernel %.3f ms (%.0f TF) |%s' % (H, W, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, var), flush=True)
print('all levels: library %.3f ms, own %.3f ms' % tuple(tot))
