"""GPU (MI355X): soft-NMS kernels (csrc/softnms.hip) through the C-ABI against the oracle -- bit
for bit (selection order, indices, decayed scores, both methods, discards, ties) -- and against the
reference-generated fixture tests/golden/soft_nms.npz.  (The oracle is compared with the
reference's own Cython module in the build container, tests/test_oracle_softnms.py.)"""
import os

import numpy as np
import pytest
import torch

import synth
import gpu_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from iouaware import ops as o
    return o


def _rand_dets(rs, n, span, ties):
    xy = rs.uniform(0, span, (n, 2))
    wh = rs.uniform(5, 60, (n, 2))
    s = rs.uniform(0.05, 1, n)
    if ties:
        s = np.round(s * 8) / 8
        xy = np.round(xy / 8) * 8
        wh = np.round(wh / 16) * 16 + 8
    return np.concatenate([xy, xy + wh, s[:, None]], 1).astype(np.float32)


def test_device_exp_f64_equals_oracle(ops, oracle_lib):
    rs = np.random.RandomState(4)
    ov = rs.uniform(0, 1, 300000).astype(np.float32)
    num = (-(ov * ov)).astype(np.float32)
    sig = rs.choice([0.3, 0.5, 1.0, 2.0], ov.size).astype(np.float32)
    got = ops.test_math(6, torch.from_numpy(num).cuda(), torch.from_numpy(sig).cuda()).cpu().numpy()
    want = oracle_lib.vec_exp_f64((num / sig).astype(np.float64)).astype(np.float32)
    assert G.same_bits(got, want)


def test_soft_nms_op_golden(ops, golden_dir):
    from iouaware import nms_op
    f = np.load(os.path.join(golden_dir, 'soft_nms.npz'))
    for i in range(int(f['num_cases'])):
        thr, sigma, ms = [float(v) for v in f['cfg_%d' % i]]
        nd, inds = nms_op.soft_nms(torch.from_numpy(f['dets_%d' % i]).cuda(), thr,
                                   method=str(f['method_%d' % i]), sigma=sigma, min_score=ms)
        assert inds.dtype == torch.long and nd.is_cuda
        assert np.array_equal(inds.cpu().numpy(), f['inds_%d' % i]), 'case %d' % i
        assert G.same_bits(nd.cpu().numpy(), f['new_dets_%d' % i]), 'case %d' % i
    nd, inds = nms_op.soft_nms(torch.from_numpy(f['ties_dets']).cuda(), 0.3, min_score=0.05)
    assert np.array_equal(inds.cpu().numpy(), f['ties_inds'])
    assert G.same_bits(nd.cpu().numpy(), f['ties_new_dets'])
    # numpy in -> numpy out (nms_wrapper.py:76-78), CPU tensor in -> CPU tensor out
    nd, inds = nms_op.soft_nms(f['dets_4'], 0.3, min_score=0.05)
    assert isinstance(nd, np.ndarray) and inds.dtype == np.int64
    assert np.array_equal(inds, f['inds_4'])
    nd, inds = nms_op.soft_nms(torch.from_numpy(f['dets_4']), 0.3, min_score=0.05)
    assert not nd.is_cuda and np.array_equal(inds.numpy(), f['inds_4'])
    e, ei = nms_op.soft_nms(torch.zeros(0, 5).cuda(), 0.3)
    assert e.shape == (0, 5) and ei.shape == (0,)


def test_soft_nms_op_random_vs_oracle(ops, oracle_lib):
    """HIP vs the oracle on random / tie-heavy cases.  The oracle itself is compared with the
    reference's own Cython module in the BUILD CONTAINER only (tests/test_oracle_softnms.py);
    on the GPU box the reference is represented by the committed fixture soft_nms.npz above."""
    rs = np.random.RandomState(6)
    sizes = [1, 2, 63, 64, 65, 255, 256, 257, 700, 2100, 4693]
    for trial, n in enumerate(sizes * 2):
        d = _rand_dets(rs, n, float(rs.choice([60, 200, 700])), ties=(trial % 2 == 1))
        for method, code in (('linear', 1), ('gaussian', 2)):
            thr = float(rs.choice([0.3, 0.5]))
            sigma = float(rs.choice([0.3, 0.5, 1.0]))
            ms = float(rs.choice([1e-3, 0.05, 0.3]))
            nd, inds = ops.soft_nms_dets(torch.from_numpy(d).cuda(), thr, method, sigma, ms)
            ob, oi = oracle_lib.soft_nms(d, thr, method, sigma, ms)
            assert np.array_equal(inds.cpu().numpy(), oi), (n, method, trial)
            assert G.same_bits(nd.cpu().numpy(), ob), (n, method, trial)


def _soft_oracle(oracle_lib, cls, reg, iou, b, base, img_hw, sf, nms_pre, score_thr, kw, mp):
    pre = oracle_lib.get_bboxes_single([x[b] for x in cls], [x[b] for x in reg],
                                       [x[b] for x in iou], synth.STRIDES, base, img_hw, sf, True,
                                       nms_pre, score_thr, 0.5, mp)
    return pre, oracle_lib.multiclass_soft_nms(pre['mlvl_bboxes'], pre['mlvl_scores'], score_thr,
                                               max_per_img=mp, **kw)


@pytest.mark.parametrize('kind', ['A', 'B', 'C'])
def test_get_bboxes_soft_vs_oracle(ops, oracle_lib, kind):
    ph, pw, B = 128, 160, 2
    cls, reg, iou = synth.head_outputs(77, B, ph, pw, kind)
    geom, base = G.geometry(ph, pw, 300)
    metas = [synth.img_meta(120, 157, ph, pw, 1.0), synth.img_meta(120, 157, ph, pw, 1.6)]
    for kw in (dict(iou_thr=0.5, method='linear', sigma=0.5, min_score=0.05),
               dict(iou_thr=0.3, method='gaussian', sigma=0.5, min_score=0.1)):
        soft = {k: v for k, v in kw.items() if k != 'iou_thr'}
        dets, labels, rows, num, dbg = ops.get_bboxes(
            geom, G.to_dev(cls), G.to_dev(reg), G.to_dev(iou), [m['img_shape'] for m in metas],
            [m['scale_factor'] for m in metas], True, 0.05, kw['iou_thr'], 100, debug=True,
            soft=soft)
        for b in range(B):
            pre, r = _soft_oracle(oracle_lib, cls, reg, iou, b, base, (120, 157),
                                  metas[b]['scale_factor'], 300, 0.05, kw, 100)
            k = int(num[b])
            assert k == r['det_bboxes'].shape[0]
            kc = dbg['keep_count'][b].cpu().numpy()
            assert np.array_equal(kc, r['keep_count'])
            kr = dbg['keep_rows'][b].cpu().numpy()
            for c in range(synth.C):
                assert np.array_equal(kr[c, :kc[c]], r['keep_rows'][c, :kc[c]]), (kind, b, c)
            assert np.array_equal(labels[b, :k].cpu().numpy(), r['det_labels'])
            assert np.array_equal(rows[b, :k].cpu().numpy(), r['det_rows'])
            assert G.same_bits(dets[b, :k].cpu().numpy(), r['det_bboxes'])


def test_head_get_bboxes_soft_nms_golden(golden_dir):
    """test_cfg.nms.type='soft_nms' through the head API, against the reference's own output"""
    import iouaware
    from iouaware.config import ConfigDict
    f = np.load(os.path.join(golden_dir, 'soft_nms.npz'))
    ih, iw, ph, pw = [int(v) for v in f['gb_img']]
    cls, reg, iou = synth.head_outputs(int(f['gb_seed']), 2, ph, pw, 'A')
    head = iouaware.build_head(dict(
        type='IoUawareRetinaHead', num_classes=81, in_channels=256, stacked_convs=4,
        feat_channels=256, octave_base_scale=4, scales_per_octave=3, anchor_ratios=[0.5, 1.0, 2.0],
        anchor_strides=[8, 16, 32, 64, 128], target_means=[.0] * 4, target_stds=[1.0] * 4,
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
        loss_bbox=dict(type='SmoothL1Loss', beta=0.11, loss_weight=1.0))).cuda()
    metas = [synth.img_meta(ih, iw, ph, pw, 1.0), synth.img_meta(ih, iw, ph, pw, 1.6)]
    for v, nms in enumerate((dict(type='soft_nms', iou_thr=0.5, min_score=0.05),
                             dict(type='soft_nms', iou_thr=0.3, method='gaussian', sigma=0.5,
                                  min_score=0.05))):
        cfg = ConfigDict(dict(nms_pre=300, min_bbox_size=0, score_thr=0.05, nms=nms,
                              max_per_img=100))
        res = head.get_bboxes(G.to_dev(cls), G.to_dev(reg), G.to_dev(iou), None, None, metas, cfg,
                              True)
        for b, (dets, labels) in enumerate(res):
            assert labels.dtype == torch.long
            assert np.array_equal(labels.cpu().numpy(), f['gb_labels_%d_%d' % (v, b)])
            assert G.close(dets.cpu().numpy(), f['gb_dets_%d_%d' % (v, b)])


def test_multiclass_nms_wrapper_soft(oracle_lib):
    from iouaware import nms_op
    rs = np.random.RandomState(12)
    n, Cn = 400, 6
    boxes = _rand_dets(rs, n, 150, False)[:, :4]
    scores = rs.uniform(0, 1, (n, Cn + 1)).astype(np.float32) ** 3
    cfg = dict(type='soft_nms', iou_thr=0.3, method='linear', min_score=0.05)
    d, l = nms_op.multiclass_nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(),
                                 0.05, cfg, 50)
    r = oracle_lib.multiclass_soft_nms(boxes, scores[:, 1:], 0.05, 0.3, 'linear', 0.5, 0.05, 50)
    assert np.array_equal(l.cpu().numpy(), r['det_labels'])
    assert G.same_bits(d.cpu().numpy(), r['det_bboxes'])


def test_get_bboxes_soft_beyond_the_batched_capacity(ops, oracle_lib):
    """test_cfg.nms.type = 'soft_nms' with more candidates per image than the batched entry holds
    (nms_pre = 2000 on a 800 x 928 pyramid: 8 259): the stage entries + one soft-NMS per class
    (ops._get_bboxes_per_class) against the oracle, like test_get_bboxes_soft_vs_oracle"""
    from iouaware import _lib
    ph, pw, B, nms_pre = 800, 928, 1, 2000
    cls, reg, iou = synth.head_outputs(78, B, ph, pw, 'C')
    geom, base = G.geometry(ph, pw, nms_pre)
    assert geom.R > _lib.IA_MAX_CANDIDATES
    kw = dict(iou_thr=0.5, method='linear', sigma=0.5, min_score=0.05)
    soft = {k: v for k, v in kw.items() if k != 'iou_thr'}
    dets, labels, rows, num, dbg = ops.get_bboxes(geom, G.to_dev(cls), G.to_dev(reg), G.to_dev(iou), [(797, 925, 3)],
                                                  [1.0], True, 0.05, kw['iou_thr'], 100, debug=True, soft=soft)
    pre, r = _soft_oracle(oracle_lib, cls, reg, iou, 0, base, (797, 925), 1.0, nms_pre, 0.05, kw, 100)
    k = int(num[0])
    assert k == r['det_bboxes'].shape[0] and k > 0
    kc = dbg['keep_count'][0].cpu().numpy()
    assert np.array_equal(kc, r['keep_count'])
    kr = dbg['keep_rows'][0].cpu().numpy()
    for c in range(synth.C):
        assert np.array_equal(kr[c, :kc[c]], r['keep_rows'][c, :kc[c]]), c
    assert np.array_equal(labels[0, :k].cpu().numpy(), r['det_labels'])
    assert np.array_equal(rows[0, :k].cpu().numpy(), r['det_rows'])
    assert G.same_bits(dets[0, :k].cpu().numpy(), r['det_bboxes'])
