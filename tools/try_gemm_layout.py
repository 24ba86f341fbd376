"""Scratch: the 1x1-convolution GEMMs of the R-50 step with the weight stored (k, n) [the inference
layout, an NN product for the library] against (n, k) [read through the transpose flag: TN], every
supporting library kernel timed for both."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
import torch
from iouaware import ops
ops.gemm_tuning('all')
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 8
shapes = [(200, 336, 64, 256), (200, 336, 256, 64), (100, 168, 512, 128), (100, 168, 128, 512), (50, 84, 1024, 256),
          (50, 84, 256, 1024), (25, 42, 2048, 512), (25, 42, 512, 2048), (100, 168, 512, 256), (25, 42, 2048, 256)]
tot = [0.0, 0.0]
for (h, w, k, n) in shapes:
    x = torch.randn(B, k, h, w, device='cuda').contiguous(memory_format=torch.channels_last)
    wkn = torch.randn(k, n, device='cuda') * 0.05
    wnk = wkn.t().contiguous()
    b = torch.randn(n, device='cuda')
    t0 = timeit(lambda: ops.linear_bias_act(x, wkn, b, relu=True))
    t1 = timeit(lambda: ops.linear_bias_act(x, wnk, b, relu=True, w_nk=True))
    fl = 2.0 * B * h * w * k * n
    tot[0] += t0; tot[1] += t1
    print('rows %6d k %4d n %4d   (k,n) %7.1f us %6.1f TF/s   (n,k) %7.1f us %6.1f TF/s' % (B * h * w, k, n, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6))
print('sum (k,n) %.1f us   (n,k) %.1f us' % tuple(tot))
