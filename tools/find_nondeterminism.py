#!/usr/bin/env python
"""Which operators do not reproduce their own bits?  Every leaf module (and every fused module of
fuse_inference) of a network is called three times on the inputs it saw during one forward; an
operator whose outputs differ between the calls is reported with its shape.

    python tools/find_nondeterminism.py [--net r50] [--path winograd] [--deterministic] [--find]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from iouaware import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--net', default='r50')
ap.add_argument('--path', default='winograd')
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--size', default='800x1344')
ap.add_argument('--deterministic', action='store_true')
ap.add_argument('--find', action='store_true')
args = ap.parse_args()
torch.backends.cudnn.benchmark = args.find
torch.backends.cudnn.deterministic = args.deterministic
dev = torch.device('cuda', 0)
NETS = {'r50': ({}, torch.float32), 'r101': (dict(depth=101), torch.float32),
        'x101-64x4d': (dict(type='ResNeXt', depth=101, groups=64, base_width=4), torch.float32),
        'r101-bf16': (dict(depth=101), torch.bfloat16)}
bb, dt = NETS[args.net]
H, W = [int(v) for v in args.size.split('x')]
m = bench.build_model(dev, fuse=args.path != 'module', channels_last=args.path == 'winograd', backbone=bb)
if dt != torch.float32:
    m = m.to(dt)
x = torch.randn(args.batch, 3, H, W, device=dev).to(dt)
if args.path == 'winograd':
    x = x.contiguous(memory_format=torch.channels_last)

saved = []


def hook(mod, inp, out):
    saved.append((mod, inp))


names = {}
handles = []
for name, mod in m.named_modules():
    names[mod] = name
    fused = hasattr(mod, '_ia_opts')
    leaf = len(list(mod.children())) == 0
    if fused or (leaf and isinstance(mod, (torch.nn.Conv2d, torch.nn.BatchNorm2d, torch.nn.MaxPool2d))):
        handles.append(mod.register_forward_hook(hook))
with torch.no_grad():
    m.forward_head(x)
torch.cuda.synchronize()
for h in handles:                 # the replay below must not record again
    h.remove()
saved = list(saved)
print('%d operator calls recorded (%s, %s, batch %d, %dx%d, deterministic=%s, find=%s)'
      % (len(saved), args.net, args.path, args.batch, H, W, args.deterministic, args.find))


def flat(o):
    if isinstance(o, torch.Tensor):
        return [o]
    out = []
    for v in o:
        out += flat(v)
    return out


bad = 0
seen = set()
for mod, inp in saved:
    kids_fused = any(hasattr(c, '_ia_opts') for c in mod.modules() if c is not mod)
    with torch.no_grad():
        outs = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            outs.append([t.clone() for t in flat(mod(*inp))])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
    same = all(torch.equal(a, b) for o in outs[1:] for a, b in zip(outs[0], o))
    shp = tuple(flat(inp)[0].shape)
    key = (type(mod).__name__, shp, getattr(mod, 'kernel_size', None), getattr(mod, 'stride', None))
    if not same:
        bad += 1
        d = max(float((a.float() - b.float()).abs().max()) for o in outs[1:] for a, b in zip(outs[0], o))
        print('NOT REPRODUCIBLE  %-40s %-14s in %s k=%s s=%s  max diff %.2e  (%.3f ms)%s'
              % (names[mod], type(mod).__name__, shp, getattr(mod, 'kernel_size', ''), getattr(mod, 'stride', ''),
                 d, ms, '  [contains fused children]' if kids_fused else ''))
    elif isinstance(mod, torch.nn.Conv2d) and key not in seen:
        print('ok                %-40s %-14s in %s k=%s s=%s (%.3f ms)' % (names[mod], type(mod).__name__, shp,
                                                                          mod.kernel_size, mod.stride, ms))
    seen.add(key)
print('%d of %d operator calls are not reproducible' % (bad, len(saved)))
print('gemm table', ops.gemm_table_stats())
