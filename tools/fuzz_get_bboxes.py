#!/usr/bin/env python
"""Randomised whole-path check: ops.get_bboxes (both head layouts, complete and lazy NMS) against the
C oracle, every stage bit for bit (tests/test_gpu_parity.py::check_against_oracle), over random
pyramid sizes / batches / nms_pre / thresholds / score statistics.  One-off hunting tool; run on an
MI355X:   python tools/fuzz_get_bboxes.py [cases] [first seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tests', 'oracle', 'iou-aware-single-stage-object-detector_amd', ''):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gpu_util as G  # noqa: E402
import synth  # noqa: E402
import oracle  # noqa: E402
from iouaware import ops  # noqa: E402
import test_gpu_parity as P  # noqa: E402

oracle.build()


def run_case(seed):
    """one random configuration (drawn from `seed`) through check_against_oracle; raises on the
    first mismatch; -> the configuration as text"""
    rs = np.random.RandomState(seed)
    big = rs.rand() < 0.15
    ph = 32 * int(rs.randint(2, 26 if big else 12))
    pw = 32 * int(rs.randint(2, 43 if big else 16))
    B = int(rs.randint(1, 3 if big else 6))
    nms_pre = int(rs.choice([17, 100, 333, 1000, 1000, 2000, 4096]))
    kind = str(rs.choice(['A', 'B', 'C'] if min(ph, pw) >= 128 else ['A', 'B']))
    score_thr = float(rs.choice([0.01, 0.05, 0.05, 0.3, 0.6]))
    iou_thr = float(rs.choice([0.3, 0.5, 0.5, 0.7]))
    max_per_img = int(rs.choice([1, 10, 100, 100, 300]))
    rescale = bool(rs.rand() < 0.7)
    dtype = torch.bfloat16 if rs.rand() < 0.2 else torch.float32
    ih, iw = ph - int(rs.randint(0, 32)), pw - int(rs.randint(0, 32))
    sf = float(rs.choice([1.0, 1.0, 0.75, 1.6666666]))
    cls, reg, iou = synth.head_outputs(seed, B, ph, pw, kind)
    if dtype == torch.bfloat16:
        cls, reg, iou = G.bf16_round(cls), G.bf16_round(reg), G.bf16_round(iou)
    geom, base = G.geometry(ph, pw, nms_pre)
    metas = [synth.img_meta(ih, iw, ph, pw, sf) for _ in range(B)]
    tag = 'seed %d: %dx%d B=%d nms_pre=%d kind=%s thr=%.2f iou=%.1f max=%d rescale=%d %s' % (
        seed, ph, pw, B, nms_pre, kind, score_thr, iou_thr, max_per_img, rescale, str(dtype)[6:])
    try:
        P.check_against_oracle(ops, oracle, cls, reg, iou, geom, base, metas, rescale, score_thr, iou_thr,
                               max_per_img, dtype=dtype)
    except Exception as exc:
        raise AssertionError(tag + ' -> %s: %s' % (type(exc).__name__, str(exc)[:300])) from exc
    return tag


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    bad = 0
    for i in range(cases):
        try:
            print('ok   ' + run_case(seed0 + i), flush=True)
        except AssertionError as exc:                   # keep hunting
            bad += 1
            print('FAIL ' + str(exc), flush=True)
    print('%d cases, %d failures, %.0f s' % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
