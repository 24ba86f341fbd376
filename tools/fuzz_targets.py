#!/usr/bin/env python
"""Randomised check of the device target assignment (ops.anchor_targets, csrc/assign.hip) against the
torch restatement of the reference's anchor_target / MaxIoUAssigner (iouaware/targets.py, pinned on
the reference's fixtures by tests/test_gpu_targets.py) over random pad sizes, batches, gt counts
(1 ... 300), tiny / border-hugging boxes and partially valid feature maps.  One-off hunting tool:
    python tools/fuzz_targets.py [cases] [first seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tests', 'oracle', 'iou-aware-single-stage-object-detector_amd', ''):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
from iouaware import ops  # noqa: E402
from iouaware.head import IoUawareRetinaHead  # noqa: E402
from iouaware.targets import anchor_target  # noqa: E402
from test_host_targets import HEAD_KW, TRAIN_CFG  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
head = IoUawareRetinaHead(**HEAD_KW)
bad = 0
t0 = time.time()
for i in range(cases):
    rs = np.random.RandomState(seed0 + i)
    ph, pw = 32 * int(rs.randint(3, 26)), 32 * int(rs.randint(3, 43))
    B = int(rs.randint(1, 6))
    gmax = int(rs.choice([1, 2, 8, 40, 300]))
    gts, gls = [], []
    for b in range(B):
        g = int(rs.randint(1, gmax + 1))
        cx, cy = rs.uniform(0, pw, g), rs.uniform(0, ph, g)
        bw = np.exp(rs.uniform(np.log(2.0), np.log(0.9 * pw), g))
        bh = np.exp(rs.uniform(np.log(2.0), np.log(0.9 * ph), g))
        bx = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
        bx[:, 0::2] = np.clip(bx[:, 0::2], 0, pw - 1)
        bx[:, 1::2] = np.clip(bx[:, 1::2], 0, ph - 1)
        gts.append(bx.astype(np.float32))
        gls.append(rs.randint(1, 81, g).astype(np.int64))
    metas = [synth.img_meta(ph - int(rs.randint(0, 31)), pw - int(rs.randint(0, 31)),
                            ph - 32 * int(rs.randint(0, 2)), pw - 32 * int(rs.randint(0, 3))) for b in range(B)]
    tag = 'case %d seed %d: %dx%d B=%d gts=%s' % (i, seed0 + i, ph, pw, B, [len(x) for x in gts])
    try:
        sizes = synth.level_shapes(ph, pw)
        geom = head.geometry(sizes, -1)
        gtb = [torch.from_numpy(x).cuda() for x in gts]
        gtl = [torch.from_numpy(x).cuda() for x in gls]
        anchors, flags = head.get_anchors(sizes, metas, device='cuda')
        ref = anchor_target(anchors, flags, gtb, metas, head.target_means, head.target_stds, TRAIN_CFG,
                            gt_labels_list=gtl, label_channels=80, sampling=False)
        labels, lw, bt, bw_, counts = ops.anchor_targets(geom, gtb, gtl, [m['pad_shape'] for m in metas],
                                                         0.5, 0.4, 0.0, -1)
        assert int(counts[:, 0].clamp(min=1).sum()) == ref[4], 'positives'
        assert int(counts[:, 1].clamp(min=1).sum()) == ref[5], 'negatives'
        for l in range(5):
            assert torch.equal(labels[l], ref[0][l].reshape(labels[l].shape)), 'labels level %d' % l
            assert torch.equal(lw[l], ref[1][l].reshape(lw[l].shape)), 'label weights level %d' % l
            assert torch.equal(bw_[l], ref[3][l].reshape(bw_[l].shape)), 'bbox weights level %d' % l
            assert torch.allclose(bt[l], ref[2][l].reshape(bt[l].shape), rtol=1e-5, atol=1e-6), 'bbox targets level %d' % l
        print('ok   ' + tag, flush=True)
    except Exception as exc:
        bad += 1
        print('FAIL ' + tag + ' -> %s: %s' % (type(exc).__name__, str(exc)[:300]), flush=True)
print('%d cases, %d failures, %.0f s' % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
