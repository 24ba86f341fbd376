#!/usr/bin/env python
"""EXPERIMENT: does the decode stage run faster right behind the kernels that WROTE the logits (Infinity Cache, 256 MiB)?
One stage pass between two events, preceded by (a) nothing (the previous pass = reads only), (b) a 1 GiB flush,
(c) an in-place rewrite of the P3 class logits (387 MB, ascending addresses), (d) a rewrite of all levels, P3 last,
(e) a rewrite of only the second half of P3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch
import bench  # noqa
import gpu_util as G
import synth
from iouaware import ops
B = 8
g = torch.Generator(device='cuda').manual_seed(5)
cls, reg, iou = [], [], []
for (h, w) in synth.level_shapes(800, 1344):
    cls.append((-4.595 + torch.randn(B, 720, h, w, device='cuda', generator=g) * 0.0016).contiguous(memory_format=torch.channels_last))
    iou.append((torch.randn(B, 9, h, w, device='cuda', generator=g) * 0.0019).contiguous(memory_format=torch.channels_last))
    reg.append((torch.randn(B, 36, h, w, device='cuda', generator=g) * 0.5).contiguous(memory_format=torch.channels_last))
geom0, _ = G.geometry(800, 1344, 1000)
geom = ops.geometry_for(geom0, cls, reg, iou)
st = ops.DecodeStage(geom, cls, reg, iou, [(800, 1333, 3)] * B, [1.0] * B, True)
flush = torch.empty(1 << 28, device='cuda')
half = cls[0].permute(0, 2, 3, 1)[B // 2:]           # images 4..7 of P3: the upper half of the buffer


def pre_none(): pass
def pre_flush(): flush.fill_(1.0)
def pre_p3(): cls[0].mul_(1.0)
def pre_all():
    for t in reversed(cls): t.mul_(1.0)
def pre_half(): half.mul_(1.0)


for name, pre in (('nothing before', pre_none), ('1 GiB flush before', pre_flush), ('P3 logits rewritten before', pre_p3),
                  ('all logits rewritten, P3 last', pre_all), ('upper half of P3 rewritten', pre_half)):
    ts = []
    for it in range(12):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); st.run(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    print('%-34s stage by events: median %.1f us  min %.1f  max %.1f' % (name, ts[len(ts) // 2], ts[0], ts[-1]))
