"""ResNet stage-1 1x1 convolutions (batch 8, 200x336): the streaming MFMA kernel with the weights in
LDS (csrc/conv1x1_stream.hip) against the library GEMM with the fused epilogue (ops.linear_bias_act)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
import torch
from iouaware import ops
ops.gemm_tuning('all')
B, H, W = 8, 200, 336
def bench(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cl = torch.channels_last
for k, n, res in ((64, 256, True), (256, 64, False), (64, 64, False), (64, 256, False)):
    x = torch.randn(B, k, H, W, device='cuda').contiguous(memory_format=cl)
    w = torch.randn(k, n, device='cuda') * 0.05
    b = torch.randn(n, device='cuda')
    r = torch.randn(B, n, H, W, device='cuda').contiguous(memory_format=cl) if res else None
    ops.STREAM_1X1 = False                       # the library GEMM, not the routing of linear_bias_act
    want = ops.linear_bias_act(x, w, b, residual=r, relu=True)
    got = ops.conv1x1_stream(x, w, b, residual=r, relu=True)
    err = float((got - want).abs().max()) / float(want.abs().max())
    mb = (x.numel() + (r.numel() if res else 0) + want.numel()) * 4 / 1e6
    t0 = bench(lambda: ops.linear_bias_act(x, w, b, residual=r, relu=True))
    t1 = bench(lambda: ops.conv1x1_stream(x, w, b, residual=r, relu=True))
    print('%3d -> %3d res=%d  %.0f MB  library %.1f us (%.2f TB/s)   own %.1f us (%.2f TB/s)   max rel diff %.1e'
          % (k, n, res, mb, t0 * 1e3, mb / t0 / 1e3, t1 * 1e3, mb / t1 / 1e3, err), flush=True)

# the block boundary in one kernel (k_conv1x1_chain) against the two kernels it replaces
x = torch.randn(B, 64, H, W, device='cuda').contiguous(memory_format=cl)
w = torch.randn(64, 256, device='cuda') * 0.05
b = torch.randn(256, device='cuda')
w2 = torch.randn(256, 64, device='cuda') * 0.05
b2 = torch.randn(64, device='cuda')
r = torch.randn(B, 256, H, W, device='cuda').contiguous(memory_format=cl)
def two():
    y = ops.conv1x1_stream(x, w, b, residual=r, relu=True)
    return y, ops.conv1x1_stream(y, w2, b2, relu=True)
y0, h0 = two()
y1, h1 = ops.conv1x1_chain(x, w, b, r, w2, b2)
t0, t1 = bench(two), bench(lambda: ops.conv1x1_chain(x, w, b, r, w2, b2))
mb = (x.numel() + r.numel() + y0.numel() + h0.numel()) * 4 / 1e6
print('boundary 64 -> 256 (+res) -> 64: two kernels %.1f us, chained %.1f us (%.0f MB, %.2f TB/s); same bits: %s'
      % (t0 * 1e3, t1 * 1e3, mb, mb / t1 / 1e3, torch.equal(y0, y1) and torch.equal(h0, h1)), flush=True)

# stage-2 tail: 128 -> 512 + residual + ReLU at 100 x 168 (k_conv1x1_wide) against the library GEMM
x = torch.randn(B, 128, 100, 168, device='cuda').contiguous(memory_format=cl)
w = torch.randn(128, 512, device='cuda') * 0.05
b = torch.randn(512, device='cuda')
r = torch.randn(B, 512, 100, 168, device='cuda').contiguous(memory_format=cl)
ops.WIDE_1X1 = False
t0 = bench(lambda: ops.linear_bias_act(x, w, b, residual=r, relu=True))
ops.WIDE_1X1 = True
t1 = bench(lambda: ops.linear_bias_act(x, w, b, residual=r, relu=True))
mb = (x.numel() + 2 * r.numel()) * 4 / 1e6
print('128 -> 512 res=1  %.0f MB  library %.1f us (%.2f TB/s)   own (wide) %.1f us (%.2f TB/s)' % (mb, t0 * 1e3, mb / t0 / 1e3, t1 * 1e3, mb / t1 / 1e3), flush=True)
