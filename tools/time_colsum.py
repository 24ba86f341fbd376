"""Scratch: k_relu_bwd_colsum (ReLU backward + bias gradient in one pass) on the activation shapes of
the R-50 training iteration, against eager's threshold_backward + sum and the HBM time of
3 floats per element (2 without the mask)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
import torch
from iouaware import winograd_train as WT

def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [(4, 128, 100, 168), (4, 512, 100, 168), (4, 256, 50, 84), (4, 1024, 50, 84), (4, 512, 25, 42),
          (4, 2048, 25, 42), (4, 256, 100, 168), (4, 720, 100, 168), (4, 256, 25, 42)]
for shp in shapes:
    dy = torch.randn(shp, device='cuda').contiguous(memory_format=torch.channels_last)
    y = torch.randn(shp, device='cuda').relu().contiguous(memory_format=torch.channels_last)
    for relu in (True, False):
        yy = y if relu else None
        t_k = timeit(lambda: WT.relu_bwd_bias_grad(dy, yy, True))
        def eager():
            g = torch.ops.aten.threshold_backward(dy, y, 0) if relu else dy
            return g, g.sum((0, 2, 3))
        t_e = timeit(eager)
        nbytes = dy.numel() * 4 * (3 if relu else 1)
        print('%-22s relu=%d  kernel %7.1f us (%.2f TB/s)   eager %7.1f us' % (shp, relu, t_k, nbytes / t_k / 1e6, t_e))
