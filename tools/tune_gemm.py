#!/usr/bin/env python
"""OFFLINE tuning of the library GEMM kernels (run on an MI355X; the product never times anything).

    python tools/tune_gemm.py [--out PATH] [--mode all|heuristic] [--only NAME,...]

For every workload below the networks run once with `ops.gemm_tuning(mode)`: the first call of a
GEMM shape times the library's kernels for it (csrc/gemm.hip) and the winner's solution index is
recorded.  The result -- (m, n, k, flags, batch, dtype) -> hipBLASLt solution index, plus the
library version the indices belong to -- is written as JSON and committed as
iou-aware-single-stage-object-detector_amd/iouaware/tuning/hipblaslt_gfx950.json, which
iouaware.ops loads at the first GEMM.  At run time (tests, bench.py, serving) the mode is 'frozen':
table entry or the heuristic's first result, no timing -- the same bits in every run."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import bench  # noqa: E402
from iouaware import ops  # noqa: E402

R101 = dict(depth=101)
X64 = dict(type='ResNeXt', depth=101, groups=64, base_width=4)
X32 = dict(type='ResNeXt', depth=101, groups=32, base_width=4)
# name, backbone, dtype, [(batch, H, W)]
INFER = [
    ('r50', {}, torch.float32, [(8, 800, 1344), (1, 800, 1344), (1, 256, 320), (2, 256, 320), (4, 256, 320)]),
    ('r101', R101, torch.float32, [(8, 800, 1344), (1, 800, 1344), (1, 256, 320), (2, 256, 320), (4, 256, 320)]),
    ('x101-64x4d', X64, torch.float32, [(8, 800, 1344), (1, 800, 1344), (1, 256, 320), (2, 256, 320)]),
    ('x101-32x4d', X32, torch.float32, [(1, 256, 320), (2, 256, 320)]),
    ('r101-bf16', R101, torch.bfloat16, [(16, 800, 1344), (1, 800, 1344), (4, 256, 320)]),
    ('r50-bf16', {}, torch.bfloat16, [(16, 800, 1344)]),
]


def infer(name, backbone, dtype, shapes, dev):
    model = bench.build_model(dev, fuse=True, channels_last=True, backbone=backbone)
    if dtype != torch.float32:
        model = model.to(dtype)
    for B, H, W in shapes:
        x = torch.randn(B, 3, H, W, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
        t0 = time.time()
        with torch.no_grad():
            model.forward_head(x)
        torch.cuda.synchronize()
        print('  %-12s %2d x %4d x %4d  %.1f s, %d shapes so far' % (name, B, H, W, time.time() - t0,
                                                                    len(ops.gemm_table_dump())), flush=True)
        del x
    del model
    torch.cuda.empty_cache()


def train(dev):
    import synth
    from iouaware.train import train_step
    model, opt, img, ms, gtb, gtl = bench.train_state(dev)
    clip = dict(max_norm=35, norm_type=2)
    t0 = time.time()
    train_step(model, opt, img, ms, gtb, gtl, grad_clip=clip)
    torch.cuda.synchronize()
    print('  r50-train     4 x  800 x 1344  %.1f s, %d shapes so far' % (time.time() - t0, len(ops.gemm_table_dump())),
          flush=True)
    # the small training shapes of the tests / smoke()
    img2 = torch.randn(2, 3, 256, 320, device=dev).contiguous(memory_format=torch.channels_last)
    gts, gls = synth.train_targets(11, 2, 250, 317, max_gt=5)
    meta = [dict(ori_shape=(250, 317, 3), img_shape=(250, 317, 3), pad_shape=(256, 320, 3),
                 scale_factor=1.0, flip=False)] * 2
    train_step(model, opt, img2, meta, [torch.from_numpy(g).to(dev) for g in gts],
               [torch.from_numpy(g).to(dev) for g in gls], grad_clip=clip)
    torch.cuda.synchronize()
    del model, opt
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'hipblaslt_gfx950.json'))
    ap.add_argument('--mode', default='all', choices=['all', 'heuristic'])
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.backends.cudnn.benchmark = False
    ops.gemm_table_load(path='/nonexistent')          # start from an empty table
    ops.gemm_tuning(args.mode)
    only = set(filter(None, args.only.split(',')))
    t0 = time.time()
    for name, bb, dt, shapes in INFER:
        if only and name not in only:
            continue
        infer(name, bb, dt, shapes, dev)
    if not only or 'r50-train' in only:
        train(dev)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    n = ops.gemm_table_save(args.out, merge=bool(only))
    try:                                   # the rest of this gpurun call uses it
        import shutil
        shutil.copyfile(args.out, ops.GEMM_TABLE)
    except OSError:
        pass
    print('wrote %s: %d shapes, hipBLASLt version %d, %.0f s' % (args.out, n, ops._lib.lib().ia_gemm_library_version(),
                                                                time.time() - t0))


if __name__ == '__main__':
    main()
