"""GPU (MI355X): the HIP path, called through the C-ABI, against the CPU oracle
on the same seeded inputs and against the reference-generated golden fixtures.

Bars
  * HIP vs oracle: BIT-EXACT for every float and every index (the kernels and
    the oracle restate the same IEEE operation sequence), on all inputs
    including heavy ties.
  * HIP vs golden (the reference itself): indices bit-exact, floats
    |a-b| <= 1e-4 * max(1,|b|)  (north_star tolerance 1e-4 fp32).
"""
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


# ------------------------------------------------------------------ device math
def test_device_math_is_bit_identical_to_oracle(ops, oracle_lib):
    rs = np.random.RandomState(3)
    x = np.concatenate([
        rs.standard_normal(400000) * 8, rs.uniform(-104, 89, 200000), rs.uniform(-1e-3, 1e-3, 1000),
        [0.0, -0.0, 1.0, -1.0, 88.72, 88.73, -87.3, -103.9, -104.0, 4.1351666, -4.1351666,
         1e-38, -1e-38, 1e-45, 16.6, -16.6, 17.0, -17.0]]).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    for op, name in ((0, 'expf'), (2, 'sigmoidf')):
        assert G.same_bits(ops.test_math(op, xd).cpu().numpy(), oracle_lib.vec(name, x)), name
    pos = np.abs(x) + np.float32(1e-30)
    assert G.same_bits(ops.test_math(1, torch.from_numpy(pos).cuda()).cpu().numpy(),
                       oracle_lib.vec('logf', pos))
    # correctly rounded sqrt and divide (hipcc default) == numpy fp32
    assert G.same_bits(ops.test_math(3, torch.from_numpy(pos).cuda()).cpu().numpy(), np.sqrt(pos))
    y = (rs.uniform(0.1, 1000, x.size)).astype(np.float32)
    assert G.same_bits(ops.test_math(4, xd, torch.from_numpy(y).cuda()).cpu().numpy(), x / y)
    sq = np.sqrt(oracle_lib.vec('sigmoidf', x))
    assert G.same_bits(ops.test_math(5, xd).cpu().numpy(), sq)


# ------------------------------------------------------------------ stages
def oracle_image(oracle_lib, cls, reg, iou, b, base, img_hw, sf, rescale, nms_pre, score_thr,
                 iou_thr, max_per_img, means=(0, 0, 0, 0), stds=(1, 1, 1, 1), softmax=False,
                 C_cls=synth.C):
    return oracle_lib.get_bboxes_single([x[b] for x in cls], [x[b] for x in reg],
                                        [x[b] for x in iou], synth.STRIDES, base, img_hw, sf,
                                        rescale, nms_pre, score_thr, iou_thr, max_per_img,
                                        means=means, stds=stds, softmax=softmax, C_cls=C_cls)


def check_against_oracle(ops, oracle_lib, cls, reg, iou, geom, base, metas, rescale, score_thr,
                         iou_thr, max_per_img, dtype=torch.float32, means=(0, 0, 0, 0),
                         stds=(1, 1, 1, 1), layouts=(True, False)):
    """both memory orders of the head outputs: NCHW (k_rowmax) and channels-last, consumed in
    place (k_rowmax_nhwc) -- every stage bit for bit against the oracle"""
    for channels_last in layouts:
        out = _check_layout(ops, oracle_lib, cls, reg, iou, geom, base, metas, rescale, score_thr,
                            iou_thr, max_per_img, dtype, means, stds, channels_last)
    return out


def _check_layout(ops, oracle_lib, cls, reg, iou, geom, base, metas, rescale, score_thr, iou_thr,
                  max_per_img, dtype, means, stds, channels_last):
    B = cls[0].shape[0]
    dc, dr, di = G.to_dev(cls, dtype), G.to_dev(reg, dtype), G.to_dev(iou, dtype)
    if channels_last:
        dc, dr, di = [[t.contiguous(memory_format=torch.channels_last) for t in x]
                      for x in (dc, dr, di)]
        # consumed in place when a class row is a whole number of 16-byte vectors (81 softmax
        # channels are not: those heads are transposed to NCHW by ops.level_ptrs)
        in_place = (geom.Cin * dc[0].element_size()) % 16 == 0
        assert ops.geometry_for(geom, dc, dr, di).layout == (1 if in_place else 0)
    natural = channels_last and (geom.Cin * dc[0].element_size()) % 16 == 0     # row-max array in p*A + a order
    shapes = [m['img_shape'] for m in metas]
    sfs = [m['scale_factor'] for m in metas]
    dets, labels, rows, num, dbg = ops.get_bboxes(geom, dc, dr, di, shapes, sfs, rescale,
                                                  score_thr, iou_thr, max_per_img, debug=True)
    torch.cuda.synchronize()
    # the lazy evaluation of the NMS (default product path) returns the very same tensors; with 128
    # / 100 candidates most images run out of candidates and take the gated complete path
    for cand in (0, 128, max_per_img):
        lz = ops.get_bboxes(geom, dc, dr, di, shapes, sfs, rescale, score_thr, iou_thr, max_per_img,
                            lazy=True, lazy_candidates=cand)
        for name, x, y in zip(('dets', 'labels', 'rows', 'num'), lz, (dets, labels, rows, num)):
            assert torch.equal(x, y), 'lazy (%d candidates) %s differs' % (cand, name)
    dets, labels, rows, num = dets.cpu().numpy(), labels.cpu().numpy(), rows.cpu().numpy(), \
        num.cpu().numpy()
    dbg = {k: v.cpu().numpy() for k, v in dbg.items()}
    out = []
    for b in range(B):
        o = oracle_image(oracle_lib, cls, reg, iou, b, base, shapes[b][:2], sfs[b], rescale,
                         geom.struct.nms_pre, score_thr, iou_thr, max_per_img, means, stds,
                         softmax=geom.softmax, C_cls=geom.C)
        # device layout for NCHW heads: per level an (A, HW) block; channels-last heads and the
        # oracle: reference order p*A + a
        off = 0
        for (h, w) in geom.featmap_sizes:
            n_l = h * w * geom.A
            dev = dbg['rowmax'][b][off:off + n_l]
            if not natural:
                dev = dev.reshape(geom.A, h * w).T.reshape(-1)
            assert G.same_bits(dev, o['rowmax'][off:off + n_l]), 'rowmax img %d' % b
            off += n_l
        assert np.array_equal(dbg['cand_idx'][b], o['topk_inds']), 'topk img %d' % b
        assert G.same_bits(dbg['boxes'][b], o['mlvl_bboxes']), 'boxes img %d' % b
        assert G.same_bits(dbg['scores_t'][b][:, :geom.R].T, o['mlvl_scores']), 'scores img %d' % b
        assert np.array_equal(dbg['keep_count'][b], o['keep_count']), 'keep_count img %d' % b
        for c in range(geom.C):
            k = o['keep_count'][c]
            assert np.array_equal(dbg['keep_rows'][b, c, :k], o['keep_rows'][c, :k]), (b, c)
        n = int(num[b])
        assert n == o['num_det']
        assert G.same_bits(dets[b, :n], o['det_bboxes'])
        assert np.array_equal(labels[b, :n], o['det_labels'])
        assert np.array_equal(rows[b, :n], o['det_rows'])
        assert (labels[b, n:] == -1).all() and (dets[b, n:] == 0).all()
        out.append(dict(det_bboxes=dets[b, :n], det_labels=labels[b, :n], det_rows=rows[b, :n],
                        topk_inds=dbg['cand_idx'][b], keep_count=dbg['keep_count'][b],
                        keep_rows=np.concatenate([dbg['keep_rows'][b, c, :dbg['keep_count'][b, c]]
                                                  for c in range(geom.C)])))
    return out


@pytest.mark.parametrize('name', ['small', 'dense', 'full_A', 'full_C', 'vecscale', 'softmax'])
def test_get_bboxes_vs_oracle_and_golden(ops, oracle_lib, golden_dir, name):
    """'softmax': the use_sigmoid_cls=False branch (iou_aware_retina_head.py:506-507,540-541), 81
    class channels per anchor, fixture from the reference head built with a softmax loss"""
    f = np.load(os.path.join(golden_dir, 'get_bboxes_%s.npz' % name))
    ih, iw, ph, pw = [int(v) for v in f['img']]
    B = int(f['batch'])
    from test_oracle_golden import scale_factor_of, fixture_inputs
    cls, reg, iou = fixture_inputs(f)
    assert synth.checksum(cls + reg + iou) == int(f['checksum'])
    geom, base = G.geometry(ph, pw, int(f['nms_pre']), softmax=name == 'softmax')
    # 'vecscale': the 4-vector scale_factor of a non-keep-ratio resize (transforms.py:33-38)
    metas = [synth.img_meta(ih, iw, ph, pw, scale_factor_of(f, b)) for b in range(B)]
    res = check_against_oracle(ops, oracle_lib, cls, reg, iou, geom, base, metas,
                               bool(f['rescale']), float(f['score_thr']), float(f['iou_thr']),
                               int(f['max_per_img']))
    for b, r in enumerate(res):     # and against the reference's own outputs
        assert np.array_equal(r['topk_inds'], f['topk_inds_%d' % b])
        assert np.array_equal(r['keep_count'], f['keep_count_%d' % b])
        assert np.array_equal(r['keep_rows'], f['keep_rows_%d' % b])
        assert np.array_equal(r['det_labels'], f['det_labels_%d' % b])
        assert np.array_equal(r['det_rows'], f['det_rows_%d' % b])
        assert G.close(r['det_bboxes'], f['det_bboxes_%d' % b], 1e-4)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_softmax_head_channels_last_in_place(ops, oracle_lib, dtype):
    """softmax branch with 79 foreground classes: 80 class channels per anchor = whole 16-byte
    vectors, so channels-last head outputs are consumed in place (the NHWC arm of
    k_rowscore_softmax / k_gather_softmax); every stage bit for bit against the oracle, fp32 and
    bf16 storage, plus the stage-wise C-ABI entries (grouped row-max -> top-k)"""
    from iouaware import ops as iops
    ph, pw, B, Cf = 320, 384, 2, 79          # P3: 40 x 48 x 9 = 17 280 anchors: a FILTERED level of the top-k
    rs = np.random.RandomState(4242)
    cls, reg, iou = [], [], []
    for (h, w) in synth.level_shapes(ph, pw):
        c = (rs.standard_normal((B, synth.A, Cf + 1, h, w)) * 2.5).astype(np.float32)
        c[:, :, 0] += np.float32(2.0)
        cls.append(np.ascontiguousarray(c.reshape(B, synth.A * (Cf + 1), h, w)))
        reg.append((rs.standard_normal((B, synth.A * 4, h, w)) * 0.5).astype(np.float32))
        iou.append((rs.standard_normal((B, synth.A, h, w)) * 1.5).astype(np.float32))
    if dtype == torch.bfloat16:
        cls, reg, iou = G.bf16_round(cls), G.bf16_round(reg), G.bf16_round(iou)
    _, base = G.geometry(ph, pw, 300)
    geom = iops.HeadGeometry(synth.level_shapes(ph, pw), synth.STRIDES, base, Cf, nms_pre=300,
                             softmax=True)
    assert geom.Cin == 80 and geom.softmax
    metas = [synth.img_meta(310, 377, ph, pw, 1.0), synth.img_meta(310, 377, ph, pw, 1.6)]
    check_against_oracle(ops, oracle_lib, cls, reg, iou, geom, base, metas, True, 0.05, 0.5, 100,
                         dtype=dtype)
    # stage entries: row scores (+ group maxima derived behind them) -> top-k, both orders
    for channels_last in (True, False):
        dc, dr, di = G.to_dev(cls, dtype), G.to_dev(reg, dtype), G.to_dev(iou, dtype)
        if channels_last:
            dc, dr, di = [[t.contiguous(memory_format=torch.channels_last) for t in x]
                          for x in (dc, dr, di)]
        g = iops.geometry_for(geom, dc, dr, di)
        assert g.layout == (1 if channels_last else 0)
        rowmax = iops.decode_fuse_rowmax(g, dc, dr, di)
        cand = iops.select_topk(g, rowmax)
        ws = iops.select_workspace(g, B, dc[0].device).zero_()
        cand_grouped = iops.select_topk(g, iops.decode_fuse_rowmax(g, dc, dr, di, ws), ws)
        assert torch.equal(cand, cand_grouped)
        for b in range(B):
            o = oracle_image(oracle_lib, cls, reg, iou, b, base, (310, 377), 1.0, True, 300, 0.05,
                             0.5, 100, softmax=True, C_cls=Cf)
            assert np.array_equal(cand[b].cpu().numpy(), o['topk_inds'])


def test_softmax_head_module_matches_reference(golden_dir):
    """IoUawareRetinaHead built with a softmax classification loss (use_sigmoid_cls=False):
    81 class channels per anchor, get_bboxes against the reference head's own detections"""
    from iouaware.head import IoUawareRetinaHead
    from iouaware.config import ConfigDict
    f = np.load(os.path.join(golden_dir, 'get_bboxes_softmax.npz'))
    ih, iw, ph, pw = [int(v) for v in f['img']]
    head = IoUawareRetinaHead(num_classes=81, in_channels=256, stacked_convs=4, feat_channels=256,
                              octave_base_scale=4, scales_per_octave=3, anchor_ratios=[0.5, 1.0, 2.0],
                              anchor_strides=[8, 16, 32, 64, 128],
                              loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                              loss_bbox=dict(type='SmoothL1Loss', beta=0.11, loss_weight=1.0))
    assert not head.use_sigmoid_cls and head.cls_out_channels == 81
    assert head.retina_cls.out_channels == 9 * 81
    cls, reg, iou = synth.head_outputs_softmax(int(f['seed']), int(f['batch']), ph, pw)
    cfg = ConfigDict(dict(nms_pre=int(f['nms_pre']), min_bbox_size=0, score_thr=float(f['score_thr']),
                          nms=dict(type='nms', iou_thr=float(f['iou_thr'])), max_per_img=int(f['max_per_img'])))
    metas = [synth.img_meta(ih, iw, ph, pw, float(s)) for s in f['scale_factors']]
    for channels_last in (False, True):
        dev = [G.to_dev(x) for x in (cls, reg, iou)]
        if channels_last:
            dev = [[t.contiguous(memory_format=torch.channels_last) for t in x] for x in dev]
        res = head.get_bboxes(dev[0], dev[1], dev[2], None, None, metas, cfg, True)
        for b, (dets, labels) in enumerate(res):
            assert labels.dtype == torch.int64
            assert np.array_equal(labels.cpu().numpy(), f['det_labels_%d' % b])
            assert G.close(dets.cpu().numpy(), f['det_bboxes_%d' % b], 1e-4)


def test_heavy_ties_random_init_like(ops, oracle_lib):
    """BASELINE config 1: random-init weights give almost constant logits, every fused score
    in [0.0704, 0.0710] with rampant fp32 ties.  Canonical order (score desc,
    index asc) must hold bit for bit, including ties AT the k boundary."""
    ph, pw, B = 128, 160, 2
    rs = np.random.RandomState(8)
    cls, reg, iou = [], [], []
    for (h, w) in synth.level_shapes(ph, pw):
        # few distinct values -> massive ties
        cls.append((-4.595 + rs.randint(-3, 4, (B, 720, h, w)) * 0.0016).astype(np.float32))
        reg.append((rs.standard_normal((B, 36, h, w)) * 0.01).astype(np.float32))
        iou.append((rs.randint(-2, 3, (B, 9, h, w)) * 0.0019).astype(np.float32))
    geom, base = G.geometry(ph, pw, 1000)
    metas = [synth.img_meta(120, 157, ph, pw, 1.0) for _ in range(B)]
    check_against_oracle(ops, oracle_lib, cls, reg, iou, geom, base, metas, True, 0.05, 0.5, 100)
    # all logits identical: every anchor ties
    cls = [np.full_like(c, -4.595) for c in cls]
    iou = [np.zeros_like(i) for i in iou]
    check_against_oracle(ops, oracle_lib, cls, reg, iou, geom, base, metas, True, 0.05, 0.5, 100)


def test_target_stds_and_no_rescale(ops, oracle_lib):
    ph, pw = 96, 128
    cls, reg, iou = synth.head_outputs(77, 1, ph, pw, 'B')
    stds = (0.1, 0.1, 0.2, 0.2)
    means = (0.01, -0.02, 0.03, 0.0)
    geom, base = G.geometry(ph, pw, 500, means, stds)
    metas = [synth.img_meta(90, 121, ph, pw, 1.0)]
    check_against_oracle(ops, oracle_lib, cls, reg, iou, geom, base, metas, False, 0.1, 0.45, 64,
                         means=means, stds=stds)


def test_nothing_passes_score_thr(ops, oracle_lib):
    ph, pw = 64, 64
    cls, reg, iou = synth.head_outputs(5, 1, ph, pw, 'A')
    cls = [c - 30.0 for c in cls]
    geom, base = G.geometry(ph, pw, 1000)
    metas = [synth.img_meta(64, 64, ph, pw, 1.0)]
    res = check_against_oracle(ops, oracle_lib, cls, reg, iou, geom, base, metas, True, 0.05, 0.5,
                               100)
    assert res[0]['det_bboxes'].shape == (0, 5)      # bbox_nms.py:57-59 empty result


def test_bf16_inputs(ops, oracle_lib):
    """config 3: bf16 feature maps, fp32 math.  The oracle is fed the same bf16-rounded values."""
    ph, pw = 128, 160
    cls, reg, iou = synth.head_outputs(31, 2, ph, pw, 'A')
    cls, reg, iou = G.bf16_round(cls), G.bf16_round(reg), G.bf16_round(iou)
    geom, base = G.geometry(ph, pw, 1000)
    metas = [synth.img_meta(120, 157, ph, pw, 1.0), synth.img_meta(120, 157, ph, pw, 2.0)]
    check_against_oracle(ops, oracle_lib, cls, reg, iou, geom, base, metas, True, 0.05, 0.5, 100,
                         dtype=torch.bfloat16)


def test_stage_entry_points(ops, oracle_lib):
    """the four stage entry points chained by hand equal the fused driver"""
    ph, pw = 128, 160
    cls, reg, iou = synth.head_outputs(9, 2, ph, pw, 'A')
    geom, base = G.geometry(ph, pw, 300)
    dc, dr, di = G.to_dev(cls), G.to_dev(reg), G.to_dev(iou)
    shapes, sfs = [(120, 157, 3)] * 2, [1.0, 1.3]
    _stage_chain(ops, geom, dc, dr, di, shapes, sfs)
    dc, dr, di = [[t.contiguous(memory_format=torch.channels_last) for t in x] for x in (dc, dr, di)]
    _stage_chain(ops, ops.geometry_for(geom, dc, dr, di), dc, dr, di, shapes, sfs)


def _stage_chain(ops, geom, dc, dr, di, shapes, sfs):
    rm = ops.decode_fuse_rowmax(geom, dc, dr, di)
    idx = ops.select_topk(geom, rm)
    boxes, scores_t, best = ops.gather_decode(geom, dc, dr, di, idx, shapes, sfs, True)
    assert torch.equal(best, scores_t[:, :, :geom.R].max(1).values)
    d1 = ops.multiclass_nms(boxes, scores_t, geom.R, 0.05, 0.5, 100, best_score=best)
    d0 = ops.multiclass_nms(boxes, scores_t, geom.R, 0.05, 0.5, 100)      # without the activity filter
    for a, b in zip(d0[:5], d1[:5]):          # dets, labels, rows, num, keep_count
        assert torch.equal(a, b)
    for cand in (0, 100):
        d3 = ops.multiclass_nms_lazy(boxes, scores_t, geom.R, 0.05, 0.5, 100, best_score=best,
                                     candidates=cand)
        for a, b in zip(d1[:4], d3):
            assert torch.equal(a, b)
    d2 = ops.get_bboxes(geom, dc, dr, di, shapes, sfs, True, 0.05, 0.5, 100)
    torch.cuda.synchronize()
    for a, b in zip(d1[:4], d2):
        assert torch.equal(a, b)
    # ia_decode_stage: the first three stages in one call, in the ia_get_bboxes workspace
    st = ops.DecodeStage(geom, dc, dr, di, shapes, sfs, True)
    st.run()
    v = st.views()
    torch.cuda.synchronize()
    assert torch.equal(v['rowmax'], rm) and torch.equal(v['cand_idx'], idx)
    assert torch.equal(v['boxes'], boxes) and torch.equal(v['best_score'], best)
    assert torch.equal(v['scores_t'][:, :, :geom.R], scores_t[:, :, :geom.R])


def _numpy_topk(rowmax_ref_order, k):
    """canonical order of the reference's topk (:536-544): score descending, anchor index
    ascending among equal scores -- a stable argsort of the negated scores"""
    return np.argsort(-rowmax_ref_order.astype(np.float64), kind='stable')[:k].astype(np.int32)


SELECT_CASES = [
    # pad_h, pad_w, batch, nms_pre, kind -- which top-k path the large levels take
    (800, 1344, 2, 1000, 'A'),       # P3: groups of 64 (N % 64 = 32: groups straddle images), P4: of 16
    (800, 1344, 2, 1000, 'D'),       # random-init-like near-ties: a handful of equal scores on every cut
    (384, 480, 3, 1000, 'A'),        # P3 = 25 920 anchors: groups of 4, 6 480 of them -> sampled
    (384, 480, 3, 1000, 'T'),        # few distinct values: thousands of ties on the cut
    (384, 480, 2, 1000, 'E'),        # all scores equal: every anchor is a candidate (global path)
    (384, 480, 2, 4096, 'A'),        # the largest nms_pre
    (416, 472, 2, 300, 'A'),         # H*W of P3 = 52 * 59 (odd): the scalar-load branch of k_rowmax
    (384, 480, 2, 1000, 'C'),        # clustered high scores: far more than k candidates per group
]


@pytest.mark.parametrize('ph,pw,B,nms_pre,kind', SELECT_CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_select_topk_filtered_paths(ops, ph, pw, B, nms_pre, kind, dtype):
    """the top-k behind the row-max kernel (group maxima -> filter -> exact select) on levels
    large enough to be filtered, both memory orders, against a stable sort of the very row-max
    array the device produced; the chained form (group maxima from the row-max kernel) and the
    stage form (derived from the array by an extra pass) must agree"""
    rs = np.random.RandomState(hash((ph, pw, kind)) % (2 ** 31))
    sizes = synth.level_shapes(ph, pw)
    cls, reg, iou = [], [], []
    for (h, w) in sizes:
        if kind in ('A', 'C'):
            mu, sd, isd, _ = synth.SETS[kind]
            c = rs.standard_normal((B, 720, h, w)) * sd + mu
            i = rs.standard_normal((B, 9, h, w)) * isd
            if kind == 'C':                  # blobs of high scores, 12 x 12 positions each
                for b in range(B):
                    for _ in range(6):
                        y, x = rs.randint(0, max(h - 12, 1)), rs.randint(0, max(w - 12, 1))
                        c[b, :, y:y + 12, x:x + 12] += 9.0
        elif kind == 'D':
            c = -4.595 + rs.standard_normal((B, 720, h, w)) * 0.0016
            i = rs.standard_normal((B, 9, h, w)) * 0.0019
        elif kind == 'T':
            c = -4.595 + rs.randint(-3, 4, (B, 720, h, w)) * 0.0016
            i = rs.randint(-2, 3, (B, 9, h, w)) * 0.0019
        else:
            c = np.full((B, 720, h, w), -4.595)
            i = np.zeros((B, 9, h, w))
        cls.append(c.astype(np.float32))
        iou.append(i.astype(np.float32))
        reg.append(np.zeros((B, 36, h, w), np.float32))
    geom0, _ = G.geometry(ph, pw, nms_pre)
    for channels_last in (True, False):
        dc, dr, di = G.to_dev(cls, dtype), G.to_dev(reg, dtype), G.to_dev(iou, dtype)
        if channels_last:
            dc, dr, di = [[t.contiguous(memory_format=torch.channels_last) for t in x]
                          for x in (dc, dr, di)]
        geom = ops.geometry_for(geom0, dc, dr, di)
        assert geom.layout == int(channels_last)
        ws = ops.select_workspace(geom, B, dc[0].device)
        rm = ops.decode_fuse_rowmax(geom, dc, dr, di, ws)
        idx = ops.select_topk(geom, rm, ws).cpu().numpy()
        rm2 = ops.decode_fuse_rowmax(geom, dc, dr, di)
        idx2 = ops.select_topk(geom, rm2).cpu().numpy()
        assert torch.equal(rm, rm2)
        assert np.array_equal(idx, idx2), 'chained and stage forms differ'
        rm = rm.cpu().numpy()
        off = coff = 0
        for (h, w) in sizes:
            n_l = h * w * geom.A
            k = min(n_l, nms_pre)
            for b in range(B):
                sc = rm[b, off:off + n_l]
                if not channels_last:            # stored anchor-major -> reference order p*A + a
                    sc = sc.reshape(geom.A, h * w).T.reshape(-1)
                want = _numpy_topk(sc, k) if k < n_l else np.arange(n_l, dtype=np.int32)
                got = idx[b, coff:coff + k]
                assert np.array_equal(got, want), (channels_last, (h, w), b,
                                                   int((got != want).sum()))
            off += n_l
            coff += k


@pytest.mark.parametrize('kind', ['A', 'D'])
def test_fused_rowmax_filter_launch_repeated_under_load(ops, kind):
    """ia_decode_stage on channels-last heads: row-max wavefronts and the top-k filter workgroups
    share ONE launch, the filter waiting for its segment's arrivals (write-through stores + agent
    acquire).  60 back-to-back launches at BASELINE's size while a second stream keeps the memory
    system busy: every launch must reproduce the separate kernels' row maxima and candidate list
    word for word, and leave the status word at 0."""
    ph, pw, B = 800, 1344, 8
    g = torch.Generator(device='cuda').manual_seed(5)
    cls, reg, iou = [], [], []
    for (h, w) in synth.level_shapes(ph, pw):
        if kind == 'D':
            cls.append(-4.595 + torch.randn(B, 720, h, w, device='cuda', generator=g) * 0.0016)
            iou.append(torch.randn(B, 9, h, w, device='cuda', generator=g) * 0.0019)
        else:
            cls.append(torch.randn(B, 720, h, w, device='cuda', generator=g) * 2 - 6)
            iou.append(torch.randn(B, 9, h, w, device='cuda', generator=g) * 1.5)
        reg.append(torch.randn(B, 36, h, w, device='cuda', generator=g) * 0.5)
    cls, reg, iou = [[t.contiguous(memory_format=torch.channels_last) for t in x] for x in (cls, reg, iou)]
    geom0, _ = G.geometry(ph, pw, 1000)
    geom = ops.geometry_for(geom0, cls, reg, iou)
    shapes, sfs = [(800, 1333, 3)] * B, [1.0] * B
    rm = ops.decode_fuse_rowmax(geom, cls, reg, iou)          # the separate kernels
    idx = ops.select_topk(geom, rm)
    boxes, scores_t, best = ops.gather_decode(geom, cls, reg, iou, idx, shapes, sfs, True)
    st = ops.DecodeStage(geom, cls, reg, iou, shapes, sfs, True)
    v = st.views()
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
    bad = torch.zeros((), dtype=torch.int64, device='cuda')
    for it in range(60):
        if it % 3 != 2:                                        # uneven load: two launches in three
            with torch.cuda.stream(side):
                junk.add_(1.0)
        st.run()
        bad += (v['cand_idx'] != idx).sum() + (v['rowmax'] != rm).sum() + (v['boxes'] != boxes).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0
    assert ops.get_bboxes_status(geom, B, st.ws) == (0, 0)
    assert torch.equal(v['scores_t'][:, :, :geom.R], scores_t[:, :, :geom.R])
    # the product entry point on its own state workspace, twice (the second call starts from the
    # state the first one left behind)
    for _ in range(2):
        d = ops.get_bboxes(geom, cls, reg, iou, shapes, sfs, True, 0.05, 0.5, 100, debug=True)
        assert torch.equal(d[4]['cand_idx'], idx) and torch.equal(d[4]['boxes'], boxes)


# ------------------------------------------------------------------ nms op
def test_nms_op_golden_cases(ops, golden_dir):
    n = np.load(os.path.join(golden_dir, 'nms.npz'))
    for i in range(int(n['num_cases'])):
        d = torch.from_numpy(n['dets_%d' % i]).cuda().view(-1, 5)
        keep = ops.nms_indices(d, float(n['thr_%d' % i])).cpu().numpy()
        assert np.array_equal(keep, n['keep_%d' % i]), 'case %d' % i
    for j in range(2):      # the ">=" corner (nms_cpu.cpp:55)
        d = torch.from_numpy(n['edge_dets_%d' % j]).cuda()
        keep = ops.nms_indices(d, float(n['edge_thr_%d' % j])).cpu().numpy()
        assert np.array_equal(keep, n['edge_keep_%d' % j])


def test_nms_op_threshold_rounding_adversarial(ops, oracle_lib):
    """pairs whose exact IoU sits within an ulp of the threshold: the fp64 mid-point test in
    the kernel must reproduce fl32(inter/union) >= thr exactly."""
    rs = np.random.RandomState(12)
    for thr in (0.5, 1.0 / 3.0, 0.3, 0.7, 0.45):
        thr32 = float(np.float32(thr))
        dets = []
        for _ in range(400):
            w, h = rs.randint(8, 200), rs.randint(8, 200)
            # shift so that inter/union is close to thr: inter = (w-s)*h, union = (w+s)*h
            s = int(round(w * (1 - thr) / (1 + thr)))
            for ds in (-1, 0, 1):
                x0, y0 = rs.randint(0, 5000) * 300.0, 0.0
                dets.append([x0, y0, x0 + w - 1, y0 + h - 1, rs.uniform(0.5, 1.0)])
                dets.append([x0 + s + ds, y0, x0 + s + ds + w - 1, y0 + h - 1, rs.uniform(0.0, 0.5)])
        dets = np.array(dets, np.float32)
        dets[:, 4] = (rs.permutation(len(dets)) + 1) / (len(dets) + 1.0)
        want = oracle_lib.nms(dets, thr32)
        got = ops.nms_indices(torch.from_numpy(dets).cuda(), thr32).cpu().numpy()
        assert np.array_equal(got, want), thr


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 127, 1025, 4693, 8192])
def test_nms_op_sizes(ops, oracle_lib, n):
    rs = np.random.RandomState(n)
    x1, y1 = rs.uniform(0, 600, n), rs.uniform(0, 600, n)
    dets = np.stack([x1, y1, x1 + rs.uniform(4, 150, n), y1 + rs.uniform(4, 150, n),
                     rs.randint(0, max(2, n // 3), n) / float(n)], 1).astype(np.float32)  # score ties
    got = ops.nms_indices(torch.from_numpy(dets).cuda(), 0.5).cpu().numpy()
    assert np.array_equal(got, oracle_lib.nms(dets, 0.5))


def test_nms_op_empty_and_limits(ops):
    from iouaware import _lib
    assert ops.nms_indices(torch.zeros(0, 5).cuda(), 0.5).numel() == 0
    with pytest.raises(_lib.IouAwareLibraryError):          # soft-NMS keeps its single-problem limit
        ops.soft_nms_dets(torch.zeros(8193, 5).cuda(), 0.5)


@pytest.mark.parametrize('n,extent,ties', [(8193, 600, True), (8192 * 2, 3000, False), (20001, 900, True),
                                           (40000, 6000, False), (30000, 250, True)])
def test_nms_op_beyond_one_chunk(ops, oracle_lib, n, extent, ties):
    """mmdet.ops.nms.nms has no size limit (nms_cpu.cpp:4-59); above IA_MAX_CANDIDATES ia_nms
    runs the same greedy NMS over chunks of sorted boxes (bignms.hip).  Dense scenes (nearly
    everything suppressed by earlier chunks), sparse scenes (kept list of several thousand
    boxes) and score ties across chunk borders, bit-exact against the oracle."""
    rs = np.random.RandomState(n)
    x1, y1 = rs.uniform(0, extent, n), rs.uniform(0, extent, n)
    sc = rs.randint(0, n // 5, n) / float(n) if ties else rs.uniform(0, 1, n)
    dets = np.stack([x1, y1, x1 + rs.uniform(4, 150, n), y1 + rs.uniform(4, 150, n), sc],
                    1).astype(np.float32)
    want = oracle_lib.nms(dets, 0.5)
    d = torch.from_numpy(dets).cuda()
    got = ops.nms_indices(d, 0.5).cpu().numpy()
    assert np.array_equal(got, want), (len(got), len(want))
    assert np.array_equal(ops.nms_indices(d, 0.5).cpu().numpy(), want)      # workspace reuse


# ------------------------------------------------------------------ full size, batch 8
def test_full_size_batch8_properties(ops, oracle_lib):
    """BASELINE config 2 geometry (batch 8 at 800x1344).  Size-independent properties:
    batch invariance (image b of the batch == the same image alone), score order, NMS
    postcondition, and image 0 against the oracle."""
    ph, pw, B = 800, 1344, 8
    geom, base = G.geometry(ph, pw, 1000)
    metas = [synth.img_meta(800, 1333, ph, pw, 1.0) for _ in range(B)]
    shapes, sfs = [m['img_shape'] for m in metas], [1.0] * B
    cls, reg, iou = synth.head_outputs(2024, B, ph, pw, 'C')
    dc, dr, di = G.to_dev(cls), G.to_dev(reg), G.to_dev(iou)
    dets, labels, rows, num = [t.cpu().numpy() for t in
                               ops.get_bboxes(geom, dc, dr, di, shapes, sfs, True, 0.05, 0.5, 100)]
    for b in (0, 3, 7):
        one = [t.cpu().numpy() for t in ops.get_bboxes(
            geom, [x[b:b + 1] for x in dc], [x[b:b + 1] for x in dr], [x[b:b + 1] for x in di],
            shapes[:1], sfs[:1], True, 0.05, 0.5, 100)]
        assert int(one[3][0]) == int(num[b])
        assert np.array_equal(one[0][0], dets[b]) and np.array_equal(one[2][0], rows[b])
    o = oracle_image(oracle_lib, cls, reg, iou, 0, base, (800, 1333), 1.0, True, 1000, 0.05, 0.5,
                     100)
    n0 = int(num[0])
    assert n0 == o['num_det'] and G.same_bits(dets[0, :n0], o['det_bboxes'])
    assert np.array_equal(rows[0, :n0], o['det_rows'])
    for b in range(B):
        n = int(num[b])
        s = dets[b, :n, 4]
        assert (s[:-1] >= s[1:]).all() and (s > 0.05).all()
        # NMS postcondition: same-class survivors overlap < thr
        for c in np.unique(labels[b, :n]):
            bb = dets[b, :n][labels[b, :n] == c, :4].astype(np.float32)
            for i in range(len(bb)):
                for j in range(i + 1, len(bb)):
                    xx1, yy1 = max(bb[i, 0], bb[j, 0]), max(bb[i, 1], bb[j, 1])
                    xx2, yy2 = min(bb[i, 2], bb[j, 2]), min(bb[i, 3], bb[j, 3])
                    inter = max(0., xx2 - xx1 + 1) * max(0., yy2 - yy1 + 1)
                    ai = (bb[i, 2] - bb[i, 0] + 1) * (bb[i, 3] - bb[i, 1] + 1)
                    aj = (bb[j, 2] - bb[j, 0] + 1) * (bb[j, 3] - bb[j, 1] + 1)
                    assert inter / (ai + aj - inter) < 0.5


@pytest.mark.parametrize('Cn,dtype', [(12, torch.float32), (4, torch.float32), (128, torch.float32),
                                      (8, torch.bfloat16), (40, torch.bfloat16)])
def test_channels_last_other_class_counts(ops, oracle_lib, Cn, dtype):
    """the run-time vectors-per-row path of k_rowmax_nhwc (C*sizeof != 320 / 160 bytes)"""
    ph, pw, B = 96, 128, 2
    sizes = synth.level_shapes(ph, pw)
    base = oracle_lib.head_base_anchors(synth.STRIDES)
    geom = ops.HeadGeometry(sizes, synth.STRIDES, base, Cn, nms_pre=200)
    rs = np.random.RandomState(Cn)
    cls = [(rs.standard_normal((B, 9 * Cn, h, w)) * 2 - 3).astype(np.float32) for (h, w) in sizes]
    reg = [(rs.standard_normal((B, 36, h, w)) * 0.5).astype(np.float32) for (h, w) in sizes]
    iou = [(rs.standard_normal((B, 9, h, w)) * 1.5).astype(np.float32) for (h, w) in sizes]
    if dtype == torch.bfloat16:
        cls, reg, iou = G.bf16_round(cls), G.bf16_round(reg), G.bf16_round(iou)
    dev = [[t.contiguous(memory_format=torch.channels_last) for t in G.to_dev(x, dtype)]
           for x in (cls, reg, iou)]
    assert ops.geometry_for(geom, *dev).layout == 1
    dets, labels, rows, num = ops.get_bboxes(geom, *dev, [(90, 125, 3)] * B, [1.0] * B, True, 0.05,
                                             0.5, 100)
    for b in range(B):
        o = oracle_lib.get_bboxes_single([x[b] for x in cls], [x[b] for x in reg],
                                         [x[b] for x in iou], synth.STRIDES, base, (90, 125), 1.0,
                                         True, 200, 0.05, 0.5, 100, C_cls=Cn)
        n = int(num[b])
        assert n == o['num_det']
        assert np.array_equal(labels[b, :n].cpu().numpy(), o['det_labels'])
        assert np.array_equal(rows[b, :n].cpu().numpy(), o['det_rows'])
        assert G.same_bits(dets[b, :n].cpu().numpy(), o['det_bboxes'])


def test_multiclass_nms_wrapper_max_num_minus_one_quirk(ops, golden_dir):
    """bbox_nms.py:52-56 with the default max_num=-1: `shape[0] > -1` is always true, ALL
    survivors are sorted by score (descending) and `inds[:-1]` drops the globally lowest one.
    Fixture: the reference's own multiclass_nms on 3 classes whose LAST class holds the highest
    scores (tests/golden/mnms_quirk.npz) -- the class-major tail is a high-score box there."""
    from iouaware import nms_op
    f = np.load(os.path.join(golden_dir, 'mnms_quirk.npz'))
    boxes = torch.from_numpy(f['boxes']).cuda()
    scores = torch.from_numpy(f['scores']).cuda()
    cfg = dict(type='nms', iou_thr=float(f['iou_thr']))
    thr = float(f['score_thr'])
    for name, mx in (('m1', -1), ('k20', 20), ('all', 1024)):
        b, l = nms_op.multiclass_nms(boxes, scores, thr, cfg, mx)
        assert l.dtype == torch.long
        assert np.array_equal(l.cpu().numpy(), f['labels_' + name]), name
        assert np.array_equal(b.cpu().numpy(), f['bboxes_' + name]), name    # gathered, not computed
    b, l = nms_op.multiclass_nms(boxes, scores, thr, cfg)                     # the default
    assert np.array_equal(b.cpu().numpy(), f['bboxes_m1'])
    # sorted by score, and the dropped one is the global minimum of the survivors
    assert (np.diff(f['bboxes_m1'][:, 4]) <= 0).all()
    assert f['bboxes_all'][:, 4].min() < f['bboxes_m1'][:, 4].min()


@pytest.mark.parametrize('ph,pw,B,nms_pre,max_per_img,kind', [
    (800, 928, 1, 2000, 100, 'C'),        # 2000 + 2000 + 2000 + 1755 + 504 = 8259 candidates per image > IA_MAX_CANDIDATES
    (416, 1248, 2, 4096, 300, 'A'),       # nms_pre at IA_MAX_NMS_PRE
    (256, 320, 2, 1000, 1500, 'B'),       # max_per_img > IA_MAX_PER_IMG
])
def test_get_bboxes_beyond_the_batched_capacity(ops, oracle_lib, ph, pw, B, nms_pre, max_per_img, kind):
    """The reference's get_bboxes / multiclass_nms take any nms_pre and any max_per_img
    (iou_aware_retina_head.py:499-564, bbox_nms.py:33-56); the batched C-ABI entry holds
    IA_MAX_CANDIDATES candidates and IA_MAX_PER_IMG detections per image.  Beyond that
    ops.get_bboxes takes the stage entries + one NMS per class (ops._get_bboxes_per_class) and must
    give the oracle's result, every stage bit for bit, in both layouts."""
    from iouaware import _lib
    geom, base = G.geometry(ph, pw, nms_pre)
    assert geom.R > _lib.IA_MAX_CANDIDATES or max_per_img > _lib.IA_MAX_PER_IMG
    cls, reg, iou = synth.head_outputs(77, B, ph, pw, kind)
    metas = [synth.img_meta(ph - 3, pw - 5, ph, pw, 1.0) for _ in range(B)]
    res = check_against_oracle(ops, oracle_lib, cls, reg, iou, geom, base, metas, True, 0.05, 0.5, max_per_img)
    assert all(len(r['det_labels']) > 0 for r in res)
