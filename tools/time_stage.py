#!/usr/bin/env python
"""decode stage (ia_decode_stage) at the benchmark's size on random-init-like head outputs: HIP
events around back-to-back passes (IA_FUSED_ROWMAX_FILTER=0: separate row-max / filter kernels).
Also the harness of the one-launch experiment, tools/experiments/decode_stage_one_launch.hip.txt."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch  # noqa: E402
import bench  # noqa: E402
import gpu_util as G  # noqa: E402
import synth  # noqa: E402
from iouaware import ops  # noqa: E402

B = int(os.environ.get('B', 8))
g = torch.Generator(device='cuda').manual_seed(5)
cls, reg, iou = [], [], []
for (h, w) in synth.level_shapes(800, 1344):
    cls.append((-4.595 + torch.randn(B, 720, h, w, device='cuda', generator=g) * 0.0016).contiguous(memory_format=torch.channels_last))
    iou.append((torch.randn(B, 9, h, w, device='cuda', generator=g) * 0.0019).contiguous(memory_format=torch.channels_last))
    reg.append((torch.randn(B, 36, h, w, device='cuda', generator=g) * 0.5).contiguous(memory_format=torch.channels_last))
geom0, _ = G.geometry(800, 1344, 1000)
geom = ops.geometry_for(geom0, cls, reg, iou)
st = ops.DecodeStage(geom, cls, reg, iou, [(800, 1333, 3)] * B, [1.0] * B, True)
for _ in range(3):
    st.run()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        st.run()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
print('%s: %.1f us per pass -> %.3f of 8 TB/s' % (' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith('IA_')) or 'default',
                                                  best * 1e3, 68544000.0 * B / (best * 1e-3) / 8e12))
