#!/usr/bin/env python
"""Is the product reproducible?  For each network: build it twice in this process, run the head
outputs three times each, and compare BITS; print a checksum so that two processes (and two GPU
boxes) can be compared by eye / by diff.

    python tools/check_determinism.py [--find] [--tune all]    # the old pick-by-timing behaviour
"""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from iouaware import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--find', action='store_true')
ap.add_argument('--tune', default='frozen')
ap.add_argument('--nets', default='r50,x101-64x4d,r101-bf16')
ap.add_argument('--paths', default='winograd,module')
args = ap.parse_args()
torch.backends.cudnn.benchmark = args.find
ops.gemm_tuning(args.tune)
dev = torch.device('cuda', 0)
NETS = {'r50': ({}, torch.float32), 'r101': (dict(depth=101), torch.float32),
        'x101-64x4d': (dict(type='ResNeXt', depth=101, groups=64, base_width=4), torch.float32),
        'r101-bf16': (dict(depth=101), torch.bfloat16)}


def digest(outs):
    h = hashlib.sha256()
    for ts in outs:
        for t in ts:
            h.update(t.float().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


for net in args.nets.split(','):
    bb, dt = NETS[net]
    for path in args.paths.split(','):
        for B, H, W in ((2, 256, 320), (2, 800, 1344)):
            sums = []
            for build in range(2):
                m = bench.build_model(dev, fuse=path != 'module', channels_last=path == 'winograd', backbone=bb)
                if dt != torch.float32:
                    m = m.to(dt)
                g = torch.Generator(device=dev).manual_seed(5)
                x = torch.randn(B, 3, H, W, device=dev, generator=g).to(dt)
                if path == 'winograd':
                    x = x.contiguous(memory_format=torch.channels_last)
                with torch.no_grad():
                    for rep in range(3):
                        sums.append(digest(m.forward_head(x)))
                del m
            print('%-11s %-8s %d x %4d x %4d  %s  %s' % (net, path, B, H, W, sums[0],
                                                         'REPRODUCIBLE (6 runs, 2 builds)' if len(set(sums)) == 1
                                                         else 'DIFFERS: %s' % sorted(set(sums))), flush=True)
print('gemm table', ops.gemm_table_stats())
