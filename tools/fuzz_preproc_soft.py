#!/usr/bin/env python
"""Randomised checks of two SURVEY 8(f) rows against the C oracle:
  * ia_image_transform (iouaware.preprocess.ImageTransform): random source sizes (1 ... 1500 px), target
    scales, keep-ratio on / off, flip -- every output float bit for bit, img_shape / pad_shape /
    scale_factor equal;
  * get_bboxes with test_cfg.nms.type = 'soft_nms': random pyramid sizes, batches, methods, thresholds
    -- keep lists, labels, rows and decayed detections bit for bit.
    python tools/fuzz_preproc_soft.py [cases] [first seed]"""
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
from iouaware.preprocess import ImageTransform  # noqa: E402

oracle.build()
NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


def preproc_case(seed):
    rs = np.random.RandomState(seed)
    h, w = int(np.exp(rs.uniform(0, np.log(1500)))), int(np.exp(rs.uniform(0, np.log(1500))))
    h, w = max(h, 1), max(w, 1)
    scale = (int(rs.randint(16, 1400)), int(rs.randint(16, 900)))
    keep, flip = bool(rs.rand() < 0.7), bool(rs.rand() < 0.5)
    div = int(rs.choice([32, 32, 1, 64]))
    tag = 'seed %d preproc %dx%d -> scale %s keep=%d flip=%d divisor=%d' % (seed, h, w, scale, keep, flip, div)
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    tf = ImageTransform(size_divisor=div, **NORM)
    try:
        want, ishape, pshape, sf = oracle.image_transform(img, scale, flip, keep, size_divisor=div, **NORM)
    except ValueError as e:                      # an empty rescaled image: the reference (cv2.resize) raises too
        try:
            tf(img, scale, flip, keep)
        except ValueError:
            return tag + '  (both reject: %s)' % e
        raise AssertionError(tag + ': the oracle rejects (%s), the HIP path does not' % e)
    got, gi, gp, gsf = tf(img, scale, flip, keep)
    assert tuple(gi) == tuple(ishape) and tuple(gp) == tuple(pshape), tag + ' shapes %s %s vs %s %s' % (gi, gp, ishape, pshape)
    assert np.array_equal(np.asarray(gsf, np.float64), np.asarray(sf, np.float64)), tag + ' scale factor'
    assert G.same_bits(got.cpu().numpy(), want), tag + ' pixels'
    return tag


def soft_case(seed):
    rs = np.random.RandomState(seed)
    ph, pw = 32 * int(rs.randint(2, 9)), 32 * int(rs.randint(2, 11))
    B = int(rs.randint(1, 4))
    kind = str(rs.choice(['A', 'B', 'C'] if min(ph, pw) >= 128 else ['A', 'B']))
    nms_pre = int(rs.choice([100, 300, 1000]))
    kw = dict(iou_thr=float(rs.choice([0.3, 0.5])), method=str(rs.choice(['linear', 'gaussian'])),
              sigma=float(rs.choice([0.5, 0.3])), min_score=float(rs.choice([0.05, 0.1, 0.001])))
    thr, mp = float(rs.choice([0.05, 0.2])), int(rs.choice([10, 100]))
    ih, iw = ph - int(rs.randint(0, 32)), pw - int(rs.randint(0, 32))
    sfs = [float(rs.choice([1.0, 1.6])) for _ in range(B)]
    tag = 'seed %d soft %dx%d B=%d nms_pre=%d kind=%s %s thr=%.2f max=%d' % (seed, ph, pw, B, nms_pre, kind, kw, thr, mp)
    cls, reg, iou = synth.head_outputs(seed, B, ph, pw, kind)
    geom, base = G.geometry(ph, pw, nms_pre)
    soft = {k: v for k, v in kw.items() if k != 'iou_thr'}
    dets, labels, rows, num, dbg = ops.get_bboxes(geom, G.to_dev(cls), G.to_dev(reg), G.to_dev(iou), [(ih, iw, 3)] * B, sfs,
                                                  True, thr, kw['iou_thr'], mp, debug=True, soft=soft)
    for b in range(B):
        pre = oracle.get_bboxes_single([x[b] for x in cls], [x[b] for x in reg], [x[b] for x in iou], synth.STRIDES,
                                       base, (ih, iw), sfs[b], True, nms_pre, thr, 0.5, mp)
        r = oracle.multiclass_soft_nms(pre['mlvl_bboxes'], pre['mlvl_scores'], thr, max_per_img=mp, **kw)
        k = int(num[b])
        assert k == r['det_bboxes'].shape[0], tag + ' count'
        kc = dbg['keep_count'][b].cpu().numpy()
        assert np.array_equal(kc, r['keep_count']), tag + ' keep_count'
        kr = dbg['keep_rows'][b].cpu().numpy()
        for c in range(synth.C):
            assert np.array_equal(kr[c, :kc[c]], r['keep_rows'][c, :kc[c]]), tag + ' keep_rows class %d' % c
        assert np.array_equal(labels[b, :k].cpu().numpy(), r['det_labels']), tag + ' labels'
        assert np.array_equal(rows[b, :k].cpu().numpy(), r['det_rows']), tag + ' rows'
        assert G.same_bits(dets[b, :k].cpu().numpy(), r['det_bboxes']), tag + ' dets'
    return tag


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 800
    bad, t0 = 0, time.time()
    for i in range(cases):
        try:
            print('ok   ' + (preproc_case if i % 2 == 0 else soft_case)(seed0 + i), flush=True)
        except Exception as exc:
            bad += 1
            print('FAIL seed %d -> %s: %s' % (seed0 + i, type(exc).__name__, str(exc)[:400]), flush=True)
    print('%d cases, %d failures, %.0f s' % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
