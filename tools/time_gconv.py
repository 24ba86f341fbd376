"""The grouped 3x3 convolutions of X-101-64x4d (BASELINE config 4) at batch 8, 800 x 1344: time per layer
shape, HBM and MFMA floors (csrc/gconv.hip).  PMC=1: the run is meant to be wrapped in rocprofv3 --pmc."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
import torch
from iouaware import ops
B = 8
cl = torch.channels_last
def bench(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [('layer1 s1', 256, 64, 200, 336, 1), ('layer2 s2', 512, 64, 200, 336, 2), ('layer2 s1', 512, 64, 100, 168, 1),
          ('layer3 s2', 1024, 64, 100, 168, 2), ('layer3 s1', 1024, 64, 50, 84, 1), ('layer4 s2', 2048, 64, 50, 84, 2),
          ('layer4 s1', 2048, 64, 25, 42, 1)]
only = os.environ.get('ONLY')
for name, C, G, H, W, s in shapes:
    if only and only not in name:
        continue
    x = torch.randn(B, C, H, W, device='cuda').contiguous(memory_format=cl)
    w = torch.randn(C, C // G, 3, 3, device='cuda') * 0.1
    b = torch.randn(C, device='cuda')
    wp = ops.pack_grouped_weight(w)
    cg = C // G
    y = ops.grouped_conv3x3(x, wp, b, G, s, True)
    ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), b.double(), stride=s, padding=1, groups=G).relu_()
    err = float((y[:1] - ref).abs().max() / ref.abs().max())
    t = bench(lambda: ops.grouped_conv3x3(x, wp, b, G, s, True), 5 if os.environ.get('PMC') else 20)
    mb = (x.numel() + y.numel()) * 4 / 1e6
    fl = y.numel() * (C // G) * 9 * 2
    pad = max(16, cg) / cg                      # zero padding of the 16 x 16 supergroup matrix
    print('%-10s C=%4d Cg=%2d %3dx%3d: %6.1f us  %.0f MB -> %.2f TB/s (floor %.0f us at 6 TB/s)   %.1f GFLOP real, x%.0f padded -> %.0f us at the MFMA peak   err %.1e'
          % (name, C, cg, H, W, t * 1e3, mb, mb / t / 1e3, mb / 6.0, fl / 1e9, pad, fl * pad / 157.3e6, err), flush=True)
