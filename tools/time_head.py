"""Scratch timing of the head-path kernels at full size (not the bench contract)."""
import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import synth, gpu_util as G
from iouaware import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kind = sys.argv[2] if len(sys.argv) > 2 else 'A'
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ph, pw = 800, 1344
geom, base = G.geometry(ph, pw, 1000)
g = torch.Generator(device='cuda').manual_seed(0)
cls, reg, iou = [], [], []
for (h, w) in synth.level_shapes(ph, pw):
    if kind == 'D':   # degenerate random-init-like
        cls.append(-4.595 + torch.randn(B, 720, h, w, device='cuda', generator=g) * 0.0016)
        reg.append(torch.randn(B, 36, h, w, device='cuda', generator=g) * 0.01)
        iou.append(torch.randn(B, 9, h, w, device='cuda', generator=g) * 0.0019)
    else:
        mu, sd, isd, rsd = synth.SETS[kind]
        cls.append(torch.randn(B, 720, h, w, device='cuda', generator=g) * sd + mu)
        reg.append(torch.randn(B, 36, h, w, device='cuda', generator=g) * rsd)
        iou.append(torch.randn(B, 9, h, w, device='cuda', generator=g) * isd)
if os.environ.get('CL', '1') != '0':      # channels-last head outputs (what bench.py's model produces)
    cls, reg, iou = [[t.contiguous(memory_format=torch.channels_last) for t in x]
                     for x in (cls, reg, iou)]
geom = ops.geometry_for(geom, cls, reg, iou)
print('layout', 'NHWC' if geom.layout else 'NCHW')
shapes, sfs = [(800, 1333, 3)] * B, [1.0] * B
SEL_WS = ops.select_workspace(geom, B, cls[0].device)     # the chained form ia_get_bboxes uses
def stage_times():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    ev[0].record(); rm = ops.decode_fuse_rowmax(geom, cls, reg, iou, SEL_WS)
    ev[1].record(); idx = ops.select_topk(geom, rm, SEL_WS)
    ev[2].record(); boxes, st, best = ops.gather_decode(geom, cls, reg, iou, idx, shapes, sfs, True)
    ev[3].record(); out = ops.multiclass_nms(boxes, st, geom.R, 0.05, 0.5, 100, best_score=best)
    ev[4].record(); torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(4)], out
for _ in range(3): stage_times()
acc = np.zeros(4)
for _ in range(iters):
    t, out = stage_times(); acc += t
acc /= iters
print('B=%d kind=%s  rowmax %.3f ms  select %.3f ms  gather %.3f ms  nms+final %.3f ms' % (B, kind, *acc))
bytes_ = 68544000 * B
print('rowmax: %.1f GB/s algorithmic (%.1f%% of 8 TB/s)' % (bytes_ / acc[0] / 1e6, bytes_ / acc[0] / 1e6 / 80))
print('kept per image', out[4].sum(1).tolist()[:4], 'num', out[3].tolist()[:4])
torch.cuda.synchronize(); t = time.time()
for _ in range(iters): ops.get_bboxes(geom, cls, reg, iou, shapes, sfs, True, 0.05, 0.5, 100)
torch.cuda.synchronize(); print('whole get_bboxes %.3f ms' % ((time.time() - t) / iters * 1e3))
