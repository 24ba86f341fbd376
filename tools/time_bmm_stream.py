"""The HBM-bound Winograd-domain batched GEMMs of bench config 2 on the streaming MFMA kernel
(ia_batched_gemm_stream) against the library's batched GEMM (frozen table kernels)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
import torch
from iouaware import winograd as wg
def bench(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for batch, rows, k, n in ((36, 33600, 64, 64), (36, 8400, 128, 128), (36, 11440, 256, 48)):
    v = torch.randn(batch, rows, k, device='cuda')
    u = torch.randn(batch, k, n, device='cuda') * 0.05
    o1, o2 = torch.empty(batch, rows, n, device='cuda'), torch.empty(batch, rows, n, device='cuda')
    wg.STREAM_BMM = False
    t0 = bench(lambda: wg.batched_gemm(v, u, o1))
    wg.STREAM_BMM = True
    t1 = bench(lambda: wg.batched_gemm(v, u, o2))
    mb = (v.numel() + o1.numel()) * 4 / 1e6
    print('(%d, %5d, %3d, %3d)  %.0f MB  library %.1f us (%.2f TB/s, %.0f TF)   own %.1f us (%.2f TB/s, %.0f TF)   max rel diff %.1e'
          % (batch, rows, k, n, mb, t0 * 1e3, mb / t0 / 1e3, 2 * batch * rows * k * n / t0 / 1e9, t1 * 1e3, mb / t1 / 1e3,
             2 * batch * rows * k * n / t1 / 1e9, float((o1 - o2).abs().max()) / float(o1.abs().max())), flush=True)
