"""CPU: the oracle (oracle/iouaware_oracle.c) against the golden vectors that
tests/golden/make_golden.py captured from the imported reference.

Bar: every index bit-exact; floats |a-b| <= 1e-4 * max(1, |b|).
"""
import os

import numpy as np
import pytest

import synth

TOL = 1e-4


def close(a, b, tol=TOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape
    if a.size == 0:
        return True
    return bool((np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))).all())


def test_base_and_grid_anchors(oracle_lib, golden_dir):
    g = np.load(os.path.join(golden_dir, 'anchors.npz'))
    base = oracle_lib.head_base_anchors(synth.STRIDES)
    for i, s in enumerate(synth.STRIDES):
        assert np.array_equal(base[i], g['base_%d' % s])
    # SURVEY 8a I1 stride-8 literal
    assert base[0].tolist() == [[-19, -7, 26, 14], [-25, -10, 32, 17], [-32, -14, 39, 21],
                                [-12, -12, 19, 19], [-16, -16, 23, 23], [-21, -21, 28, 28],
                                [-7, -19, 14, 26], [-10, -25, 17, 32], [-14, -32, 21, 39]]
    assert np.array_equal(oracle_lib.grid_anchors(base[0], 5, 7, 8), g['grid_8_5x7'])
    assert np.array_equal(oracle_lib.grid_anchors(base[2], 3, 4, 32), g['grid_32_3x4'])
    assert np.array_equal(oracle_lib.gen_base_anchors(16, [8, 16, 32], [0.5, 1, 2]),
                          g['base_16_rpn'])


def test_host_anchor_generator_matches_reference(golden_dir):
    """iouaware/anchors.py (the generator the head really uses: head.geometry -> ia_head_geom)
    directly against the reference-generated anchors.npz (I1, I2)."""
    from iouaware.anchors import AnchorGenerator
    g = np.load(os.path.join(golden_dir, 'anchors.npz'))
    scales = np.array([2 ** (i / 3) for i in range(3)]) * 4       # iou_aware_retina_head.py:81-83
    for s in synth.STRIDES:
        gen = AnchorGenerator(s, scales, [0.5, 1.0, 2.0])
        assert gen.base_anchors.dtype.is_floating_point
        assert np.array_equal(gen.base_anchors.numpy(), g['base_%d' % s])
    assert np.array_equal(AnchorGenerator(8, scales, [0.5, 1.0, 2.0]).grid_anchors((5, 7), 8).numpy(),
                          g['grid_8_5x7'])
    assert np.array_equal(AnchorGenerator(32, scales, [0.5, 1.0, 2.0]).grid_anchors((3, 4), 32).numpy(),
                          g['grid_32_3x4'])
    assert np.array_equal(AnchorGenerator(16, [8, 16, 32], [0.5, 1.0, 2.0]).base_anchors.numpy(),
                          g['base_16_rpn'])
    # and through the head: the geometry handed to the HIP kernels carries these base anchors
    import iouaware
    from iouaware.config import ConfigDict
    from test_host_model import model_cfg
    head = iouaware.build_head(ConfigDict(model_cfg()['bbox_head']))
    for i, s in enumerate(synth.STRIDES):
        assert np.array_equal(head.anchor_generators[i].base_anchors.numpy(), g['base_%d' % s])


def test_delta2bbox(oracle_lib, golden_dir):
    d = np.load(os.path.join(golden_dir, 'delta2bbox.npz'))
    assert close(oracle_lib.delta2bbox(d['rois'], d['deltas'], max_shape=(800, 1333)),
                 d['out_clamped'])
    assert close(oracle_lib.delta2bbox(d['rois'], d['deltas']), d['out_free'])
    assert close(oracle_lib.delta2bbox(d['rois'], d['deltas'], stds=(0.1, 0.1, 0.2, 0.2),
                                       max_shape=(800, 1333)), d['out_stds'])


def test_nms_cases(oracle_lib, golden_dir):
    n = np.load(os.path.join(golden_dir, 'nms.npz'))
    for i in range(int(n['num_cases'])):
        keep = oracle_lib.nms(n['dets_%d' % i], float(n['thr_%d' % i]))
        assert np.array_equal(keep, n['keep_%d' % i]), 'case %d' % i
    # nms_cpu.cpp:55 uses ">=": IoU exactly 1/3 is suppressed at thr=fp32(1/3), kept at 0.34
    assert oracle_lib.nms(n['edge_dets_0'], float(n['edge_thr_0'])).tolist() == [0]
    assert oracle_lib.nms(n['edge_dets_1'], float(n['edge_thr_1'])).tolist() == [0, 1]
    assert np.array_equal(n['edge_keep_0'], [0]) and np.array_equal(n['edge_keep_1'], [0, 1])
    # the fp32 instantiation on the threshold-representability cases of nms_f64.npz
    f = np.load(os.path.join(golden_dir, 'nms_f64.npz'))
    for j in range(3):
        keep = oracle_lib.nms(f['edge2_dets_%d' % j].astype(np.float32), float(f['edge2_thr_%d' % j]))
        assert np.array_equal(keep, f['edge2_keep_%d_float32' % j]), j


def test_nms_matches_real_reference_binary(oracle_lib):
    """oracle/_ref/nms_cpu_ref.so is the reference's own nms_cpu.cpp (travels to the GPU box)."""
    import build_ref
    mod = build_ref.load()
    if mod is None:
        pytest.skip('oracle/_ref not built')
    import torch
    rs = np.random.RandomState(5)
    for n in (1, 17, 300, 1500):
        x1, y1 = rs.uniform(0, 400, n), rs.uniform(0, 400, n)
        dets = np.stack([x1, y1, x1 + rs.uniform(4, 150, n), y1 + rs.uniform(4, 150, n),
                         rs.permutation(n) / n], 1).astype(np.float32)
        ref = mod.nms(torch.from_numpy(dets), 0.5).numpy()
        assert np.array_equal(oracle_lib.nms(dets, 0.5), ref)


def scale_factor_of(f, b):
    """python float (keep-ratio resize) or the fp32 4-vector of transforms.py:35-38"""
    sf = f['scale_factors'][b]
    return float(sf) if np.ndim(sf) == 0 else np.asarray(sf, np.float32)


def run_oracle(oracle_lib, f, b, cls, reg, iou):
    ih, iw, ph, pw = [int(v) for v in f['img']]
    base = oracle_lib.head_base_anchors(synth.STRIDES)
    return oracle_lib.get_bboxes_single(
        [x[b] for x in cls], [x[b] for x in reg], [x[b] for x in iou], synth.STRIDES, base,
        (ih, iw), scale_factor_of(f, b), bool(f['rescale']), int(f['nms_pre']),
        float(f['score_thr']), float(f['iou_thr']), int(f['max_per_img']),
        softmax=str(f['kind']) == 'softmax')


def fixture_inputs(f):
    """the synthetic head outputs a get_bboxes fixture was generated from"""
    ih, iw, ph, pw = [int(v) for v in f['img']]
    if str(f['kind']) == 'softmax':          # use_sigmoid_cls = False: 81 class channels per anchor
        return synth.head_outputs_softmax(int(f['seed']), int(f['batch']), ph, pw)
    return synth.head_outputs(int(f['seed']), int(f['batch']), ph, pw, str(f['kind']))


@pytest.mark.parametrize('name', ['small', 'dense', 'full_A', 'full_C', 'vecscale', 'softmax'])
def test_get_bboxes(oracle_lib, golden_dir, name):
    f = np.load(os.path.join(golden_dir, 'get_bboxes_%s.npz' % name))
    ih, iw, ph, pw = [int(v) for v in f['img']]
    B = int(f['batch'])
    cls, reg, iou = fixture_inputs(f)
    assert synth.checksum(cls + reg + iou) == int(f['checksum']), 'synthetic inputs drifted'
    for b in range(B):
        r = run_oracle(oracle_lib, f, b, cls, reg, iou)
        assert np.array_equal(r['topk_inds'], f['topk_inds_%d' % b])
        assert np.array_equal(r['keep_count'], f['keep_count_%d' % b])
        kr = np.concatenate([r['keep_rows'][c, :r['keep_count'][c]] for c in range(synth.C)])
        assert np.array_equal(kr, f['keep_rows_%d' % b])
        assert np.array_equal(r['det_labels'], f['det_labels_%d' % b])
        assert np.array_equal(r['det_rows'], f['det_rows_%d' % b])
        assert close(r['det_bboxes'], f['det_bboxes_%d' % b])
        if 'mlvl_bboxes_%d' % b in f:
            assert close(r['mlvl_bboxes'], f['mlvl_bboxes_%d' % b])
            assert close(r['mlvl_scores'], f['mlvl_scores_%d' % b])
