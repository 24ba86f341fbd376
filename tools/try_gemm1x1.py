"""Scratch: 1x1 convolutions of the backbone as plain GEMMs on the channels-last activation
(rows = B*H*W pixels), with the bias+ReLU epilogue inside the library GEMM."""
import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
import torch, torch.nn.functional as F
from iouaware import ops
torch.backends.cudnn.benchmark = True
B = 8
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
for (ci, co, H, W) in [(64, 64, 200, 336), (256, 64, 200, 336), (64, 256, 200, 336), (256, 128, 200, 336),
                       (512, 128, 100, 168), (128, 512, 100, 168), (512, 256, 100, 168), (1024, 256, 50, 84),
                       (256, 1024, 50, 84), (1024, 512, 50, 84), (2048, 512, 25, 42), (512, 2048, 25, 42)]:
    x = torch.randn(B, ci, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 1, 1, device='cuda') * 0.05
    bias = torch.randn(co, device='cuda')
    w2 = w.view(co, ci).t().contiguous()          # (Cin, Cout)
    wt = w.view(co, ci).contiguous()              # (Cout, Cin) -> F.linear
    x2 = x.permute(0, 2, 3, 1).reshape(-1, ci)
    assert x2.data_ptr() == x.data_ptr()
    def conv_ep():
        return ops.channel_affine_act_(F.conv2d(x, w), None, bias, relu=True)
    def conv():
        return F.conv2d(x, w)
    def gemm_act():
        return torch._addmm_activation(bias, x2, w2)        # relu(x2 @ w2 + bias)
    def gemm_plain():
        return torch.mm(x2, w2)
    def lin():
        return F.linear(x2, wt, bias)
    a = conv_ep().permute(0, 2, 3, 1).reshape(-1, co); b = gemm_act()
    err = float((a - b).abs().max() / a.abs().max())
    fl = 2 * B * H * W * ci * co / 1e9
    tc, tce, tg, tp, tl = bench(conv), bench(conv_ep), bench(gemm_act), bench(gemm_plain), bench(lin)
    print('%4d->%4d %3dx%3d  conv %.3f (%3.0f TF)  conv+ep %.3f  mm %.3f (%3.0f TF)  addmm_relu %.3f  linear+bias %.3f  err %.1e'
          % (ci, co, H, W, tc, fl / tc, tce, tp, fl / tp, tg, tl, err))
