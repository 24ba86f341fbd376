"""CPU: the soft-NMS restatement (oracle/iouaware_oracle_softnms.c) against
  * tests/golden/soft_nms.npz, captured from the imported reference (nms_wrapper.soft_nms over the
    reference's own Cython module; get_bboxes with test_cfg.nms.type='soft_nms'),
  * the reference module itself (oracle/_ref_local/soft_nms_cpu.so, built by oracle/build_ref.py from
    mmdet/ops/nms/src/soft_nms_cpu.pyx), on random and tie-heavy inputs: bit for bit.
"""
import os

import numpy as np
import pytest

import synth

TOL = 1e-4


def close(a, b, tol=TOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return a.size == 0 or bool((np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))).all())


def test_soft_nms_golden_cases(oracle_lib, golden_dir):
    f = np.load(os.path.join(golden_dir, 'soft_nms.npz'))
    for i in range(int(f['num_cases'])):
        thr, sigma, ms = [float(v) for v in f['cfg_%d' % i]]
        nd, inds = oracle_lib.soft_nms(f['dets_%d' % i], thr, str(f['method_%d' % i]), sigma, ms)
        assert np.array_equal(inds, f['inds_%d' % i]), 'case %d' % i
        assert np.array_equal(nd, f['new_dets_%d' % i]), 'case %d' % i      # bit-equal
    nd, inds = oracle_lib.soft_nms(f['ties_dets'], 0.3, 'linear', 0.5, 0.05)
    assert np.array_equal(inds, f['ties_inds']) and np.array_equal(nd, f['ties_new_dets'])


def test_get_bboxes_with_soft_nms_golden(oracle_lib, golden_dir):
    f = np.load(os.path.join(golden_dir, 'soft_nms.npz'))
    ih, iw, ph, pw = [int(v) for v in f['gb_img']]
    cls, reg, iou = synth.head_outputs(int(f['gb_seed']), 2, ph, pw, 'A')
    assert synth.checksum(cls + reg + iou) == int(f['gb_checksum'])
    base = oracle_lib.head_base_anchors(synth.STRIDES)
    variants = [dict(iou_thr=0.5, method='linear', sigma=0.5, min_score=0.05),
                dict(iou_thr=0.3, method='gaussian', sigma=0.5, min_score=0.05)]
    for v, kw in enumerate(variants):
        for b, sf in enumerate((1.0, 1.6)):
            pre = oracle_lib.get_bboxes_single([x[b] for x in cls], [x[b] for x in reg],
                                               [x[b] for x in iou], synth.STRIDES, base, (ih, iw),
                                               sf, True, 300, 0.05, 0.5, 100)
            r = oracle_lib.multiclass_soft_nms(pre['mlvl_bboxes'], pre['mlvl_scores'], 0.05,
                                               max_per_img=100, **kw)
            assert np.array_equal(r['det_labels'], f['gb_labels_%d_%d' % (v, b)])
            assert close(r['det_bboxes'], f['gb_dets_%d_%d' % (v, b)])


def _rand_dets(rs, n, span, ties):
    xy = rs.uniform(0, span, (n, 2))
    wh = rs.uniform(5, 60, (n, 2))
    s = rs.uniform(0.05, 1, n)
    if ties:
        s = np.round(s * 8) / 8
        xy = np.round(xy / 8) * 8
        wh = np.round(wh / 16) * 16 + 8
    return np.concatenate([xy, xy + wh, s[:, None]], 1).astype(np.float32)


def test_soft_nms_matches_real_reference_module(oracle_lib):
    import build_ref
    ref = build_ref.load_soft()
    if ref is None:
        pytest.skip('oracle/_ref_local/soft_nms_cpu.so not built (reference tree absent)')
    rs = np.random.RandomState(5)
    for trial in range(120):
        n = int(rs.randint(1, 400))
        d = _rand_dets(rs, n, float(rs.choice([50, 150, 600])), ties=(trial % 3 == 0))
        for method, code in (('linear', 1), ('gaussian', 2)):
            thr = float(rs.choice([0.3, 0.5, 0.7]))
            sigma = float(rs.choice([0.3, 0.5, 1.0]))
            ms = float(rs.choice([1e-3, 0.05, 0.2]))
            a, ai = ref.soft_nms_cpu(d.copy(), thr, method=code, sigma=sigma, min_score=ms)
            b, bi = oracle_lib.soft_nms(d, thr, method, sigma, ms)
            assert np.array_equal(ai, bi), (trial, method)
            assert np.array_equal(a.astype(np.float32), b), (trial, method)


def test_exp_f64_is_faithful(oracle_lib):
    rs = np.random.RandomState(9)
    x = np.concatenate([-rs.uniform(0, 40, 100000), rs.uniform(0, 5, 1000), [0.0, -745.5, -1e-300]])
    y = oracle_lib.vec_exp_f64(x)
    e = np.exp(x)
    ok = e > 0
    assert np.max(np.abs(y[ok] - e[ok]) / e[ok]) < 4.5e-16
    assert np.array_equal(y.astype(np.float32), e.astype(np.float32))   # what soft-NMS consumes
