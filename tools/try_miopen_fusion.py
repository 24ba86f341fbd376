"""Scratch: does MIOpen's fused conv+bias+relu (torch.ops.miopen.*) beat conv + our epilogue kernel
on the R-50 / FPN / head shapes (channels-last fp32, batch 8, 800x1344 input)?"""
import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
import torch, torch.nn.functional as F
from iouaware import ops
torch.backends.cudnn.benchmark = True
B = 8
shapes = [  # (Cin, Cout, k, stride, H, W, residual)
    (64, 64, 1, 1, 200, 336, False), (64, 64, 3, 1, 200, 336, False), (64, 256, 1, 1, 200, 336, True),
    (256, 128, 1, 1, 200, 336, False), (128, 128, 3, 2, 200, 336, False), (128, 512, 1, 1, 100, 168, True),
    (512, 256, 1, 1, 100, 168, False), (256, 256, 3, 2, 100, 168, False), (256, 1024, 1, 1, 50, 84, True),
    (1024, 512, 1, 1, 50, 84, False), (512, 512, 3, 2, 50, 84, False), (512, 2048, 1, 1, 25, 42, True),
    (256, 256, 3, 1, 100, 168, False),   # head tower conv, P3
    (256, 256, 3, 1, 50, 84, False),
]
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
for (ci, co, k, s, H, W, res) in shapes:
    x = torch.randn(B, ci, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, k, k, device='cuda').contiguous(memory_format=torch.channels_last) * 0.05
    bias = torch.randn(co, device='cuda')
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    z = torch.randn(B, co, Ho, Wo, device='cuda').contiguous(memory_format=torch.channels_last)
    def ours():
        y = F.conv2d(x, w, None, s, pad)
        return ops.channel_affine_act_(y, None, bias, residual=z if res else None, relu=True)
    def conv_only():
        return F.conv2d(x, w, None, s, pad)
    def fused():
        if res:
            return torch.ops.aten.miopen_convolution_add_relu(x, w, z, 1.0, bias, [s, s], [pad, pad], [1, 1], 1)
        return torch.ops.aten.miopen_convolution_relu(x, w, bias, [s, s], [pad, pad], [1, 1], 1)
    try:
        a = ours(); b = fused()
        err = (a - b).abs().max().item()
        tf = bench(fused)
    except Exception as e:
        err, tf = str(e)[:80], float('nan')
    print('%4d->%4d k%d s%d %3dx%3d res=%d  conv %.3f  conv+epilogue %.3f  miopen-fused %.3f ms  maxdiff %s'
          % (ci, co, k, s, H, W, res, bench(conv_only), bench(ours), tf, err))
