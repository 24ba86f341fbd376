"""Scratch: the head's two conv towers (independent chains of input transform -> 36 GEMMs ->
output transform) on two streams, so that the MFMA-bound GEMMs of one tower run beside the
HBM-bound transforms of the other, against the shipped single-stream version that batches both
towers into one launch per stage (72 GEMMs).  Batch 8, 800x1344 pyramid, 3 tower layers."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
import torch
from iouaware import winograd as wg, ops
ops.gemm_tuning('all')
B, F, NL = 8, 256, 3
sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
dev = torch.device('cuda')
plan = wg._Plan(sizes, B, dev)
T = plan.T
cl = torch.channels_last
def acts(c):
    return [torch.randn(B, c, h, w, device=dev).contiguous(memory_format=cl) for h, w in sizes]
a2, b2 = acts(2 * F), acts(2 * F)
ac, bc, ar, br = acts(F), acts(F), acts(F), acts(F)
u72 = [torch.randn(72, F, F, device=dev) * 0.05 for _ in range(NL)]
bias2 = torch.randn(2 * F, device=dev)
v72 = torch.empty(72, T, F, device=dev); m72 = torch.empty(72, T, F, device=dev)
vv = [torch.empty(36, T, F, device=dev) for _ in range(2)]
mm = [torch.empty(36, T, F, device=dev) for _ in range(2)]

def batched():
    x, y = a2, b2
    for i in range(NL):
        wg.input_transform(plan, x, 2, v72)
        wg.batched_gemm(v72, u72[i], m72)
        wg.output_transform(plan, m72, 2 * F, 2, bias2, True, [(0, 2 * F, y, 0)])
        x, y = y, x

streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def tower(k, x, y):
    for i in range(NL):
        wg.input_transform(plan, x, 1, vv[k])
        wg.batched_gemm(vv[k], u72[i][36 * k:36 * k + 36], mm[k])
        wg.output_transform(plan, mm[k], F, 1, bias2[F * k:F * k + F], True, [(0, F, y, 0)])
        x, y = y, x

def two_streams(offset=False):
    cur = torch.cuda.current_stream()
    for k, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            tower(k, (ac, ar)[k], (bc, br)[k])
    for s in streams:
        cur.wait_stream(s)

def one_stream_towers():
    tower(0, ac, bc); tower(1, ar, br)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

print('both towers batched, one stream (shipped): %.3f ms' % timeit(batched))
print('towers one after the other, one stream:     %.3f ms' % timeit(one_stream_towers))
print('towers on two streams:                      %.3f ms' % timeit(two_streams))
