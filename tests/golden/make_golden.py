"""Generate the golden fixtures in tests/golden/*.npz by running the REFERENCE
(imported read-only from /root/reference through ref_shim.py) on seeded
synthetic inputs.  Runs only in the build container:

    python tests/golden/make_golden.py

Fixtures hold seeds, expected outputs and an input checksum -- never reference
source.  Inputs are regenerated from the seed by tests/synth.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..'))
import ref_shim  # noqa: E402
import synth  # noqa: E402

ref_shim.install()
from mmdet.core.anchor import AnchorGenerator  # noqa: E402
from mmdet.core.bbox import delta2bbox, bbox2delta, bbox_overlaps  # noqa: E402
from mmdet.models.anchor_heads import iou_aware_retina_head as ref_head_mod  # noqa: E402
from mmdet.models.anchor_heads.iou_aware_retina_head import IoUawareRetinaHead  # noqa: E402
import mmdet.core.loss.losses as ref_losses  # noqa: E402

nw = sys.modules['mmdet.ops.nms.nms_wrapper']
torch.manual_seed(0)
torch.set_num_threads(8)

HEAD_KW = dict(num_classes=81, in_channels=256, stacked_convs=4, feat_channels=256,
               octave_base_scale=4, scales_per_octave=3, anchor_ratios=[0.5, 1.0, 2.0],
               anchor_strides=[8, 16, 32, 64, 128], target_means=[.0] * 4,
               target_stds=[1.0] * 4,
               loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25,
                             loss_weight=1.0),
               loss_bbox=dict(type='SmoothL1Loss', beta=0.11, loss_weight=1.0))


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024))


# ---------------------------------------------------------------- anchors (I1, I2)
def gen_anchors():
    out = {}
    scales = np.array([2 ** (i / 3) for i in range(3)]) * 4
    for s in (8, 16, 32, 64, 128):
        g = AnchorGenerator(s, scales, [0.5, 1.0, 2.0])
        out['base_%d' % s] = g.base_anchors.numpy()
    g = AnchorGenerator(8, scales, [0.5, 1.0, 2.0])
    out['grid_8_5x7'] = g.grid_anchors((5, 7), 8).numpy()
    g = AnchorGenerator(32, scales, [0.5, 1.0, 2.0])
    out['grid_32_3x4'] = g.grid_anchors((3, 4), 32).numpy()
    # a non-default generator (other scales/ratios)
    g = AnchorGenerator(16, [8, 16, 32], [0.5, 1.0, 2.0])
    out['base_16_rpn'] = g.base_anchors.numpy()
    save('anchors', **out)


# ---------------------------------------------------------------- delta2bbox (I7)
def gen_delta2bbox():
    rs = np.random.RandomState(7)
    n = 1024
    x1 = rs.uniform(-50, 1300, n)
    y1 = rs.uniform(-50, 780, n)
    rois = np.stack([x1, y1, x1 + rs.uniform(1, 600, n), y1 + rs.uniform(1, 600, n)], 1)
    rois = np.round(rois).astype(np.float32)
    deltas = (rs.standard_normal((n, 4)) * np.array([0.5, 0.5, 1.5, 1.5])).astype(np.float32)
    deltas[:8, 2:] = 10.0        # beyond the wh_ratio_clip
    deltas[8:16, 2:] = -10.0
    r, d = torch.from_numpy(rois), torch.from_numpy(deltas)
    out = dict(seed=7, rois=rois, deltas=deltas,
               out_clamped=delta2bbox(r, d, [0, 0, 0, 0], [1, 1, 1, 1], (800, 1333, 3)).numpy(),
               out_free=delta2bbox(r, d, [0, 0, 0, 0], [1, 1, 1, 1]).numpy(),
               out_stds=delta2bbox(r, d, [0., 0., 0., 0.], [0.1, 0.1, 0.2, 0.2],
                                   (800, 1333, 3)).numpy())
    save('delta2bbox', **out)


# ---------------------------------------------------------------- nms (I10, I11)
def rand_dets(rs, n, span=300.0):
    x1 = rs.uniform(0, span, n)
    y1 = rs.uniform(0, span, n)
    w = rs.uniform(5, 120, n)
    h = rs.uniform(5, 120, n)
    s = rs.permutation(n).astype(np.float64) / max(n, 1) * 0.9 + 0.05   # distinct scores
    return np.stack([x1, y1, x1 + w, y1 + h, s], 1).astype(np.float32)


def gen_nms():
    rs = np.random.RandomState(11)
    out = {}
    cases = [(0, 0.5), (1, 0.5), (2, 0.5), (63, 0.5), (64, 0.5), (65, 0.3), (200, 0.5),
             (1000, 0.7), (2500, 0.5), (4693, 0.5)]
    for i, (n, thr) in enumerate(cases):
        dets = rand_dets(rs, n, span=300.0 if n < 3000 else 900.0)
        _, inds = nw.nms(torch.from_numpy(dets), thr)
        out['dets_%d' % i] = dets
        out['thr_%d' % i] = np.float32(thr)
        out['keep_%d' % i] = inds.numpy()
    # the >= corner case: IoU exactly 1/3 (nms_cpu.cpp:55)
    d = np.array([[0, 0, 9, 9, 0.9], [5, 0, 14, 9, 0.8]], np.float32)   # inter 50, union 150
    for j, thr in enumerate((1.0 / 3.0, 0.34)):
        _, inds = nw.nms(torch.from_numpy(d), float(np.float32(thr)))
        out['edge_dets_%d' % j] = d
        out['edge_thr_%d' % j] = np.float32(thr)
        out['edge_keep_%d' % j] = inds.numpy()
    out['num_cases'] = len(cases)
    save('nms', **out)


def gen_nms_f64():
    """nms_cpu_kernel<double> (nms_cpu.cpp:63 dispatches float AND double): the reference's own
    compiled op on float64 boxes -- random sets, and the cases where the type decides: an IoU of
    exactly 1/3 against the threshold float(1/3) (fp32: suppressed, fp64: kept), coordinates that
    differ below fp32 resolution."""
    rs = np.random.RandomState(12)
    out = {}
    cases = [(1, 0.5), (2, 0.5), (65, 0.3), (200, 0.5), (1000, 0.7), (4693, 0.5), (9000, 0.6)]
    for i, (n, thr) in enumerate(cases):
        dets = rand_dets(rs, n, span=300.0 if n < 3000 else 900.0).astype(np.float64)
        dets[:, :4] += rs.uniform(-1e-9, 1e-9, (n, 4))               # below fp32 resolution
        dets[:, 4] = rs.permutation(n).astype(np.float64) / max(n, 1) * 0.9 + 0.05 + rs.uniform(0, 1e-12, n)
        _, inds = nw.nms(torch.from_numpy(dets), thr)
        assert inds.dtype == torch.int64
        out['dets_%d' % i] = dets
        out['thr_%d' % i] = np.float32(thr)
        out['keep_%d' % i] = inds.numpy()
    d = np.array([[0, 0, 9, 9, 0.9], [5, 0, 14, 9, 0.8]], np.float64)   # inter 50, union 150: IoU = 1/3
    for j, thr in enumerate((1.0 / 3.0, 0.34, 0.33)):
        for dt in (np.float64, np.float32):
            _, inds = nw.nms(torch.from_numpy(d.astype(dt)), float(np.float32(thr)))
            out['edge_keep_%d_%s' % (j, np.dtype(dt).name)] = inds.numpy()
        out['edge_thr_%d' % j] = np.float32(thr)
    out['edge_dets'] = d
    assert out['edge_keep_0_float64'].tolist() == [0, 1] and out['edge_keep_0_float32'].tolist() == [0]
    # the threshold arrives as a C float (nms_cpu.cpp:5 `const float threshold`) and is compared in
    # scalar_t: exactly representable thresholds (0.5, 0.25) against IoUs that hit them exactly --
    # both instantiations suppress --, and 0.3 (not representable: float(0.3) = 0.30000001192...)
    # against an IoU of 0.300000005: above 0.3, below float(0.3) -- the double instantiation keeps
    # the box, the float one (where the width rounds to 3 and 30 / 100 rounds up to float(0.3))
    # suppresses it.  A port that compares against double(0.3) gets the first of them wrong.
    e2 = [(np.array([[0, 0, 9, 9, 0.9], [0, 0, 4, 9, 0.8]], np.float64), 0.5),            # IoU = 50 / 100
          (np.array([[0, 0, 9, 9, 0.9], [0, 0, 4, 4, 0.8]], np.float64), 0.25),           # IoU = 25 / 100
          (np.array([[0, 0, 9, 9, 0.9], [0, 0, 2.00000005, 9, 0.8]], np.float64), 0.3)]   # IoU = 0.300000005
    for j, (dd, thr) in enumerate(e2):
        for dt in (np.float64, np.float32):
            _, inds = nw.nms(torch.from_numpy(dd.astype(dt)), float(np.float32(thr)))
            out['edge2_keep_%d_%s' % (j, np.dtype(dt).name)] = inds.numpy()
        out['edge2_dets_%d' % j] = dd
        out['edge2_thr_%d' % j] = np.float32(thr)
    assert out['edge2_keep_0_float64'].tolist() == [0] and out['edge2_keep_0_float32'].tolist() == [0]
    assert out['edge2_keep_1_float64'].tolist() == [0] and out['edge2_keep_1_float32'].tolist() == [0]
    assert out['edge2_keep_2_float64'].tolist() == [0, 1] and out['edge2_keep_2_float32'].tolist() == [0]
    out['num_cases'] = len(cases)
    print('nms_f64: kept', [len(out['keep_%d' % i]) for i in range(len(cases))],
          'edge', {k: v.tolist() for k, v in out.items() if k.startswith('edge_keep')})
    save('nms_f64', **out)


# ---------------------------------------------------------------- get_bboxes (I5..I9)
class Capture(object):
    """record what the reference computes inside get_bboxes_single"""

    def __enter__(self):
        self.topk, self.mlvl, self.nms_calls = [], [], []
        self._topk = torch.Tensor.topk
        self._mnms = ref_head_mod.multiclass_nms
        self._nms = nw.nms
        cap = self

        def topk(t, *a, **k):
            r = cap._topk(t, *a, **k)
            cap.topk.append((t.detach().clone(), r[1].clone()))
            return r

        def mnms(bboxes, scores, *a, **k):
            cap.mlvl.append((bboxes.clone(), scores.clone()))
            return cap._mnms(bboxes, scores, *a, **k)

        def nms(dets, *a, **k):
            r = cap._nms(dets, *a, **k)
            cap.nms_calls.append((dets.clone(), r[1].clone()))
            return r

        torch.Tensor.topk = topk
        ref_head_mod.multiclass_nms = mnms
        nw.nms = nms
        return self

    def __exit__(self, *exc):
        torch.Tensor.topk = self._topk
        ref_head_mod.multiclass_nms = self._mnms
        nw.nms = self._nms


def run_ref_get_bboxes(head, cls, reg, iou, metas, cfg, rescale, gts=None):
    B = cls[0].shape[0]
    t = lambda xs: [torch.from_numpy(x) for x in xs]   # noqa: E731
    if gts is None:
        gts = [torch.zeros(0, 4) for _ in range(B)]
    gl = [torch.zeros(g.shape[0], dtype=torch.long) for g in gts]
    results = []
    for b in range(B):
        sl = lambda xs: [x[b:b + 1] for x in t(xs)]    # noqa: E731
        with Capture() as cap:
            dets, labels = head.get_bboxes(sl(cls), sl(reg), sl(iou), [gts[b]], [gl[b]],
                                           [metas[b]], cfg, rescale)[0]
        mb, ms = cap.mlvl[0]
        R = mb.shape[0]
        # per-level topk (levels not exceeding nms_pre produce no call)
        topk_inds, ti = [], 0
        margins = []
        for l in range(len(cls)):
            Nl = cls[l].shape[2] * cls[l].shape[3] * synth.A
            if cfg.nms_pre > 0 and Nl > cfg.nms_pre:
                ms_l, idx = cap.topk[ti]
                ti += 1
                topk_inds.append(idx.numpy().astype(np.int32))
                srt = ms_l.sort(descending=True).values[:cfg.nms_pre + 1].double()
                margins.append(float(((srt[:-1] - srt[1:]) / srt[:-1]).min()))
            else:
                topk_inds.append(np.arange(Nl, dtype=np.int32))
        topk_inds = np.concatenate(topk_inds)
        assert topk_inds.shape[0] == R
        # per-class keep rows
        keep_count = np.zeros(synth.C, np.int32)
        keep_rows = []
        ci = 0
        for c in range(synth.C):
            mask = ms[:, c + 1] > cfg.score_thr
            if not mask.any():
                continue
            dets_c, inds = cap.nms_calls[ci]
            ci += 1
            rows = torch.nonzero(mask).squeeze(1)[inds]
            keep_count[c] = rows.numel()
            keep_rows.append(rows.numpy().astype(np.int32))
        keep_rows = np.concatenate(keep_rows) if keep_rows else np.zeros(0, np.int32)
        # rows of the final detections: match (box, score, label) back to candidates
        det_rows = np.zeros(dets.shape[0], np.int32)
        for d in range(dets.shape[0]):
            c = int(labels[d])
            hit = torch.nonzero((mb == dets[d, :4]).all(1) & (ms[:, c + 1] == dets[d, 4]))
            assert hit.numel() >= 1
            det_rows[d] = int(hit[0])
        results.append(dict(det_bboxes=dets.numpy(), det_labels=labels.numpy(),
                            det_rows=det_rows, topk_inds=topk_inds, keep_count=keep_count,
                            keep_rows=keep_rows, mlvl_bboxes=mb.numpy(),
                            mlvl_scores=ms[:, 1:].numpy(), topk_margin=np.array(margins)))
    return results


def gen_get_bboxes():
    head = IoUawareRetinaHead(**HEAD_KW)
    # --- small, everything stored
    cfg = ref_shim.to_cfg(dict(nms_pre=300, min_bbox_size=0, score_thr=0.05,
                               nms=dict(type='nms', iou_thr=0.5), max_per_img=100))
    seed, B, ih, iw, ph, pw = 101, 2, 120, 157, 128, 160
    cls, reg, iou = synth.head_outputs(seed, B, ph, pw, 'A')
    metas = [synth.img_meta(ih, iw, ph, pw, 1.0), synth.img_meta(ih, iw, ph, pw, 1.6)]
    res = run_ref_get_bboxes(head, cls, reg, iou, metas, cfg, True)
    out = dict(seed=seed, batch=B, img=np.array([ih, iw, ph, pw]), kind='A', nms_pre=300,
               score_thr=np.float32(0.05), iou_thr=np.float32(0.5), max_per_img=100,
               scale_factors=np.array([1.0, 1.6], np.float32), rescale=1,
               checksum=synth.checksum(cls + reg + iou))
    for b, r in enumerate(res):
        for k, v in r.items():
            out['%s_%d' % (k, b)] = v
        print('small img', b, 'dets', r['det_bboxes'].shape, 'into-nms',
              int((r['mlvl_scores'] > 0.05).sum()), 'topk margins', r['topk_margin'])
    save('get_bboxes_small', **out)

    # --- small, dense set, other thresholds, non-scalar behaviour: no rescale, with gt
    cfg2 = ref_shim.to_cfg(dict(nms_pre=1000, min_bbox_size=0, score_thr=0.2,
                                nms=dict(type='nms', iou_thr=0.6), max_per_img=50))
    seed = 202
    cls, reg, iou = synth.head_outputs(seed, 1, ph, pw, 'B')
    metas = [synth.img_meta(ih, iw, ph, pw, 1.0)]
    gts = [torch.tensor([[10., 12., 90., 100.], [30., 40., 150., 110.]])]
    res = run_ref_get_bboxes(head, cls, reg, iou, metas, cfg2, False, gts)[0]
    out = dict(seed=seed, batch=1, img=np.array([ih, iw, ph, pw]), kind='B', nms_pre=1000,
               score_thr=np.float32(0.2), iou_thr=np.float32(0.6), max_per_img=50,
               scale_factors=np.array([1.0], np.float32), rescale=0,
               checksum=synth.checksum(cls + reg + iou))
    for k in ('det_bboxes', 'det_labels', 'det_rows', 'topk_inds', 'keep_count', 'keep_rows',
              'topk_margin'):
        out[k + '_0'] = res[k]
    print('dense dets', res['det_bboxes'].shape, 'kept', int(res['keep_count'].sum()),
          'margins', res['topk_margin'])
    save('get_bboxes_dense', **out)

    # --- full size 800x1344 (BASELINE config 1 geometry), set A and set C
    cfg3 = ref_shim.to_cfg(dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05,
                                nms=dict(type='nms', iou_thr=0.5), max_per_img=100))
    for kind, seed in (('A', 1234), ('C', 4321)):
        cls, reg, iou = synth.head_outputs(seed, 1, 800, 1344, kind)
        metas = [synth.img_meta(800, 1333, 800, 1344, 1.0)]
        res = run_ref_get_bboxes(head, cls, reg, iou, metas, cfg3, True)[0]
        out = dict(seed=seed, batch=1, img=np.array([800, 1333, 800, 1344]), kind=kind,
                   nms_pre=1000, score_thr=np.float32(0.05), iou_thr=np.float32(0.5),
                   max_per_img=100, scale_factors=np.array([1.0], np.float32), rescale=1,
                   checksum=synth.checksum(cls + reg + iou))
        for k in ('det_bboxes', 'det_labels', 'det_rows', 'topk_inds', 'keep_count',
                  'keep_rows', 'topk_margin'):
            out[k + '_0'] = res[k]
        print('full', kind, 'dets', res['det_bboxes'].shape, 'kept',
              int(res['keep_count'].sum()), 'into-nms', int((res['mlvl_scores'] > 0.05).sum()),
              'margins', res['topk_margin'])
        save('get_bboxes_full_%s' % kind, **out)


def gen_get_bboxes_softmax():
    """use_sigmoid_cls = False (iou_aware_retina_head.py:506-507,540-541): the reference head built
    with a softmax classification loss -- cls_out_channels = 81, scores = softmax, row maximum over
    scores[:, 1:], no background padding in front of multiclass_nms (:556-558)."""
    kw = dict(HEAD_KW)
    kw['loss_cls'] = dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)
    head = IoUawareRetinaHead(**kw)
    assert not head.use_sigmoid_cls and head.cls_out_channels == 81
    cfg = ref_shim.to_cfg(dict(nms_pre=300, min_bbox_size=0, score_thr=0.05,
                               nms=dict(type='nms', iou_thr=0.5), max_per_img=100))
    B, ih, iw, ph, pw = 2, 120, 157, 128, 160
    for seed in range(707, 760):               # the first seed whose every decision margin is >= 1e-5 (printed)
        ok = True
        cls, reg, iou = synth.head_outputs_softmax(seed, B, ph, pw)
        metas = [synth.img_meta(ih, iw, ph, pw, 1.0), synth.img_meta(ih, iw, ph, pw, 1.6)]
        res = run_ref_get_bboxes(head, cls, reg, iou, metas, cfg, True)
        out = dict(seed=seed, batch=B, img=np.array([ih, iw, ph, pw]), kind='softmax', nms_pre=300,
                   score_thr=np.float32(0.05), iou_thr=np.float32(0.5), max_per_img=100,
                   scale_factors=np.array([1.0, 1.6], np.float32), rescale=1,
                   checksum=synth.checksum(cls + reg + iou))
        for b, r in enumerate(res):
            for k, v in r.items():
                out['%s_%d' % (k, b)] = v
            print('softmax img', b, 'dets', r['det_bboxes'].shape, 'into-nms',
                  int((r['mlvl_scores'] > 0.05).sum()), 'kept', int(r['keep_count'].sum()),
                  'topk margins', r['topk_margin'])
            # every decision of this fixture must be wide against a few ulp of softmax rounding
            ms64 = r['mlvl_scores'].astype(np.float64)
            thr_margin = float(np.abs(ms64 - 0.05).min() / 0.05)
            bb = r['mlvl_bboxes'].astype(np.float64)
            act = np.nonzero((ms64 > 0.05).any(1))[0]
            x1, y1, x2, y2 = [bb[act, k] for k in range(4)]
            ar = (x2 - x1 + 1) * (y2 - y1 + 1)
            ww = np.maximum(0, np.minimum(x2[:, None], x2[None]) - np.maximum(x1[:, None], x1[None]) + 1)
            ih_ = np.maximum(0, np.minimum(y2[:, None], y2[None]) - np.maximum(y1[:, None], y1[None]) + 1)
            ov = ww * ih_ / (ar[:, None] + ar[None] - ww * ih_)
            np.fill_diagonal(ov, 0.0)
            iou_margin = float(np.abs(ov - 0.5).min() / 0.5)
            print('    decision margins (relative): top-k %.1e, score_thr %.1e, IoU threshold %.1e (over %d active boxes)'
                  % (min(r['topk_margin']), thr_margin, iou_margin, len(act)))
            ok = ok and min(r['topk_margin']) >= 1e-5 and thr_margin >= 1e-5 and iou_margin >= 1e-5
        if ok:
            break
    assert ok
    save('get_bboxes_softmax', **out)


# ---------------------------------------------------------------- soft-NMS
def gen_soft_nms():
    rs = np.random.RandomState(23)
    out = {}
    cases = [(1, 0.3, 'linear', 0.5, 1e-3), (2, 0.3, 'gaussian', 0.5, 1e-3),
             (64, 0.3, 'linear', 0.5, 1e-3), (65, 0.5, 'gaussian', 0.5, 0.05),
             (300, 0.3, 'linear', 0.5, 0.05), (300, 0.3, 'gaussian', 0.3, 0.1),
             (1500, 0.5, 'linear', 0.5, 0.2), (1500, 0.5, 'gaussian', 1.0, 1e-3)]
    for i, (n, thr, method, sigma, ms) in enumerate(cases):
        dets = rand_dets(rs, n, span=200.0 if n < 1000 else 500.0)
        nd, inds = nw.soft_nms(torch.from_numpy(dets), thr, method=method, sigma=sigma,
                               min_score=ms)
        out['dets_%d' % i] = dets
        out['cfg_%d' % i] = np.array([thr, sigma, ms], np.float32)
        out['method_%d' % i] = method
        out['new_dets_%d' % i] = nd.numpy()
        out['inds_%d' % i] = inds.numpy()
        print('soft case', i, n, method, '->', nd.shape[0])
    # tie-heavy case: quantised boxes and scores (positions decide, soft_nms_cpu.pyx:52-56)
    d = rand_dets(rs, 200, span=80.0)
    d[:, :4] = np.round(d[:, :4] / 16) * 16 + np.array([0, 0, 8, 8], np.float32)
    d[:, 4] = np.round(d[:, 4] * 8) / 8 + 0.125
    nd, inds = nw.soft_nms(torch.from_numpy(d), 0.3, method='linear', min_score=0.05)
    out['ties_dets'], out['ties_new_dets'], out['ties_inds'] = d, nd.numpy(), inds.numpy()
    out['num_cases'] = len(cases)
    # whole get_bboxes with test_cfg.nms.type='soft_nms' on the 'small' synthetic head outputs
    head = IoUawareRetinaHead(**HEAD_KW)
    seed, B, ih, iw, ph, pw = 101, 2, 120, 157, 128, 160
    cls, reg, iou = synth.head_outputs(seed, B, ph, pw, 'A')
    metas = [synth.img_meta(ih, iw, ph, pw, 1.0), synth.img_meta(ih, iw, ph, pw, 1.6)]
    t = lambda xs: [torch.from_numpy(x) for x in xs]   # noqa: E731
    out['gb_seed'], out['gb_img'] = seed, np.array([ih, iw, ph, pw])
    out['gb_checksum'] = synth.checksum(cls + reg + iou)
    for v, nms in enumerate((dict(type='soft_nms', iou_thr=0.5, min_score=0.05),
                             dict(type='soft_nms', iou_thr=0.3, method='gaussian', sigma=0.5,
                                  min_score=0.05))):
        cfg = ref_shim.to_cfg(dict(nms_pre=300, min_bbox_size=0, score_thr=0.05, nms=nms,
                                   max_per_img=100))
        for b in range(B):
            sl = lambda xs: [x[b:b + 1] for x in t(xs)]    # noqa: E731
            dets, labels = head.get_bboxes(sl(cls), sl(reg), sl(iou), [torch.zeros(0, 4)],
                                           [torch.zeros(0, dtype=torch.long)], [metas[b]], cfg,
                                           True)[0]
            out['gb_dets_%d_%d' % (v, b)] = dets.numpy()
            out['gb_labels_%d_%d' % (v, b)] = labels.numpy()
            print('soft get_bboxes', v, b, dets.shape, float(dets[:, 4].min()))
    save('soft_nms', **out)


# ---------------------------------------------------------------- losses (T1..T3)
def gen_losses():
    head = IoUawareRetinaHead(**HEAD_KW)
    train_cfg = ref_shim.to_cfg(dict(
        assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0,
                      ignore_iof_thr=-1),
        allowed_border=-1, pos_weight=-1, debug=False))
    seed, B, ih, iw, ph, pw = 303, 2, 120, 157, 128, 160
    cls, reg, iou = synth.head_outputs(seed, B, ph, pw, 'A')
    gts, gls = synth.train_targets(seed + 1, B, ih, iw)
    metas = [synth.img_meta(ih, iw, ph, pw, 1.0) for _ in range(B)]
    out = dict(seed=seed, batch=B, img=np.array([ih, iw, ph, pw]), kind='A',
               checksum=synth.checksum(cls + reg + iou), gamma=np.float32(2.0),
               alpha=np.float32(0.25), beta=np.float32(0.11))
    for b in range(B):
        out['gt_bboxes_%d' % b] = gts[b]
        out['gt_labels_%d' % b] = gls[b]

    # capture the targets the reference computes (anchor_target output)
    captured = {}
    orig_at = ref_head_mod.anchor_target

    def at(*a, **k):
        r = orig_at(*a, **k)
        captured['t'] = r
        return r

    ref_head_mod.anchor_target = at
    rs = np.random.RandomState(99)
    for mode in ('attached', 'detached'):
        tc = [torch.from_numpy(x).requires_grad_(True) for x in cls]
        tr = [torch.from_numpy(x).requires_grad_(True) for x in reg]
        ti = [torch.from_numpy(x).requires_grad_(True) for x in iou]
        orig_iou_loss = ref_head_mod.weighted_iou_regression_loss
        if mode == 'detached':
            ref_head_mod.weighted_iou_regression_loss = \
                lambda p, t, w, avg_factor=None: orig_iou_loss(p, t.detach(), w,
                                                               avg_factor=avg_factor)
        losses = head.loss(tc, tr, ti, [torch.from_numpy(g) for g in gts],
                           [torch.from_numpy(g) for g in gls], metas, train_cfg)
        ref_head_mod.weighted_iou_regression_loss = orig_iou_loss
        total = sum(sum(v) for v in losses.values())
        total.backward()
        if mode == 'attached':
            for k, v in losses.items():
                out[k] = np.array([float(x) for x in v], np.float64)
            (labels, lw, bt, bw, npos, nneg, lvl_anchor) = captured['t']
            out['num_total_pos'] = npos
            out['num_total_neg'] = nneg
            for l in range(5):
                out['labels_%d' % l] = labels[l].numpy()
                out['label_weights_%d' % l] = lw[l].numpy()
                out['bbox_targets_%d' % l] = bt[l].numpy()
                out['bbox_weights_%d' % l] = bw[l].numpy()
        for l in range(5):
            for nm, tl in (('cls', tc), ('reg', tr), ('iou', ti)):
                g = tl[l].grad.numpy().reshape(-1)
                key = 'g_%s_%d' % (nm, l)
                if mode == 'attached' or nm == 'reg':
                    if key + '_idx' not in out:
                        out[key + '_idx'] = rs.choice(g.size, min(g.size, 3000),
                                                      replace=False).astype(np.int64)
                    out['%s_%s' % (key, mode)] = g[out[key + '_idx']]
                    out['%s_%s_sum' % (key, mode)] = np.float64(g.astype(np.float64).sum())
                    out['%s_%s_abs' % (key, mode)] = np.float64(np.abs(g.astype(np.float64)).sum())
    ref_head_mod.anchor_target = orig_at
    print('losses', {k: out[k] for k in ('loss_cls', 'loss_bbox', 'losses_iou')},
          'num_total_pos', out['num_total_pos'])
    save('losses_small', **out)

    # T4: the CUDA op's formula has no CPU implementation in the reference
    # (sigmoid_focal_loss.cpp:21-25 falls through for CPU tensors), so there is
    # no reference output to capture; tests pin it against a float64 restatement.




# ---------------------------------------------------------------- multiclass_nms, max_num = -1
def gen_mnms_quirk():
    """bbox_nms.py:52-56 with the default max_num=-1: `shape[0] > -1` always holds, so ALL
    survivors are sorted by score (descending) and `inds[:-1]` drops the globally lowest one.
    Three classes; the LAST class holds the highest scores, so dropping the tail of the class-major
    concatenation instead would lose a top detection."""
    from mmdet.core.post_processing.bbox_nms import multiclass_nms
    rs = np.random.RandomState(17)
    n = 300
    xy = rs.uniform(0, 400, (n, 2))
    boxes = np.concatenate([xy, xy + rs.uniform(8, 80, (n, 2))], 1).astype(np.float32)
    sc = np.zeros((n, 4), np.float32)
    sc[:, 1] = rs.uniform(0.0, 0.6, n)
    sc[:, 2] = rs.uniform(0.0, 0.7, n)
    sc[:, 3] = rs.uniform(0.4, 1.0, n)
    assert np.unique(sc[:, 1:]).size == 3 * n                       # tie-free
    out = dict(boxes=boxes, scores=sc, score_thr=np.float32(0.3), iou_thr=np.float32(0.5))
    cfg = dict(type='nms', iou_thr=0.5)
    for name, mx in (('m1', -1), ('k20', 20), ('all', 100000)):
        b, l = multiclass_nms(torch.from_numpy(boxes), torch.from_numpy(sc), 0.3, cfg, mx)
        out['bboxes_' + name] = b.numpy()
        out['labels_' + name] = l.numpy()
        print('mnms', name, b.shape, 'labels of the first 5', l[:5].tolist())
    assert out['bboxes_m1'].shape[0] == out['bboxes_all'].shape[0] - 1
    save('mnms_quirk', **out)


def gen_mnms_big():
    """multiclass_nms beyond the batched kernels' capacities (bbox_nms.py:33-56 takes any n and
    any max_num): 12 000 boxes, 3 classes, thousands of survivors; max_num = -1 (the quirk), 3 000
    and 100.  Inputs are regenerated from the seed (tests/synth-style RandomState recipe below);
    the fixture stores the outputs and a checksum of the inputs."""
    from mmdet.core.post_processing.bbox_nms import multiclass_nms
    boxes, sc = synth.mnms_big_inputs()
    out = dict(seed=23, n=boxes.shape[0], checksum=synth.checksum([boxes, sc]),
               score_thr=np.float32(0.2), iou_thr=np.float32(0.5))
    cfg = dict(type='nms', iou_thr=0.5)
    for name, mx in (('m1', -1), ('k3000', 3000), ('k100', 100)):
        b, l = multiclass_nms(torch.from_numpy(boxes), torch.from_numpy(sc), 0.2, cfg, mx)
        out['bboxes_' + name] = b.numpy()
        out['labels_' + name] = l.numpy().astype(np.int16)
        print('mnms_big', name, b.shape)
    assert out['bboxes_m1'].shape[0] > 1024
    save('mnms_big', **out)


# ---------------------------------------------------------------- get_bboxes, 4-vector scale_factor
def gen_get_bboxes_vecscale():
    """rescale=True with the 4-vector scale_factor of a non-keep-ratio resize
    (mmdet/datasets/transforms.py:33-38; divided out at iou_aware_retina_head.py:554)"""
    head = IoUawareRetinaHead(**HEAD_KW)
    cfg = ref_shim.to_cfg(dict(nms_pre=300, min_bbox_size=0, score_thr=0.05,
                               nms=dict(type='nms', iou_thr=0.5), max_per_img=100))
    seed, B, ih, iw, ph, pw = 111, 2, 120, 157, 128, 160
    cls, reg, iou = synth.head_outputs(seed, B, ph, pw, 'A')
    sfs = [np.array([1.25, 1.6, 1.25, 1.6], np.float32),
           np.array([0.8, 0.5, 0.8, 0.5], np.float32)]
    metas = [synth.img_meta(ih, iw, ph, pw, sfs[0]), synth.img_meta(ih, iw, ph, pw, sfs[1])]
    res = run_ref_get_bboxes(head, cls, reg, iou, metas, cfg, True)
    out = dict(seed=seed, batch=B, img=np.array([ih, iw, ph, pw]), kind='A', nms_pre=300,
               score_thr=np.float32(0.05), iou_thr=np.float32(0.5), max_per_img=100,
               scale_factors=np.stack(sfs), rescale=1, checksum=synth.checksum(cls + reg + iou))
    for b, r in enumerate(res):
        for k in ('det_bboxes', 'det_labels', 'det_rows', 'topk_inds', 'keep_count', 'keep_rows',
                  'mlvl_bboxes', 'mlvl_scores', 'topk_margin'):
            out['%s_%d' % (k, b)] = r[k]
        print('vecscale img', b, 'dets', r['det_bboxes'].shape, 'margins', r['topk_margin'])
    save('get_bboxes_vecscale', **out)


# ---------------------------------------------------------------- losses, mixed pad shapes (T1)
def gen_losses_mixed_pad():
    """Two images of DIFFERENT pad_shape in one batch tensor: the second image's valid_flags /
    inside_flags are partly false (anchor_head.py:135-146, anchor_generator.py:72-84,
    anchor_target.py:168-172,258-282 `unmap`), so its out-of-pad anchors get label 0 / weight 0."""
    head = IoUawareRetinaHead(**HEAD_KW)
    train_cfg = ref_shim.to_cfg(dict(
        assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0,
                      ignore_iof_thr=-1),
        allowed_border=-1, pos_weight=-1, debug=False))
    seed, B, ph, pw = 313, 2, 128, 160
    shapes = [(120, 157, 128, 160), (90, 100, 96, 128)]        # img_h, img_w, pad_h, pad_w
    cls, reg, iou = synth.head_outputs(seed, B, ph, pw, 'A')
    gts, gls = [], []
    for b, (ih, iw, _, _) in enumerate(shapes):
        g, l = synth.train_targets(seed + 1 + b, 1, ih, iw)
        gts.append(g[0])
        gls.append(l[0])
    metas = [synth.img_meta(*s, 1.0) for s in shapes]
    out = dict(seed=seed, batch=B, tensor=np.array([ph, pw]), shapes=np.array(shapes), kind='A',
               checksum=synth.checksum(cls + reg + iou))
    for b in range(B):
        out['gt_bboxes_%d' % b] = gts[b]
        out['gt_labels_%d' % b] = gls[b]
    captured = {}
    orig_at = ref_head_mod.anchor_target
    orig_ga = head.get_anchors

    def at(*a, **k):
        r = orig_at(*a, **k)
        captured['t'] = r
        return r

    def ga(*a, **k):
        r = orig_ga(*a, **k)
        # anchor_target concatenates these lists in place (anchor_target.py:52-54): keep copies
        captured['valid'] = [[f.clone() for f in fl] for fl in r[1]]
        return r

    ref_head_mod.anchor_target = at
    head.get_anchors = ga
    rs = np.random.RandomState(98)
    tc = [torch.from_numpy(x).requires_grad_(True) for x in cls]
    tr = [torch.from_numpy(x).requires_grad_(True) for x in reg]
    ti = [torch.from_numpy(x).requires_grad_(True) for x in iou]
    losses = head.loss(tc, tr, ti, [torch.from_numpy(g) for g in gts],
                       [torch.from_numpy(g) for g in gls], metas, train_cfg)
    ref_head_mod.anchor_target = orig_at
    sum(sum(v) for v in losses.values()).backward()
    for k, v in losses.items():
        out[k] = np.array([float(x) for x in v], np.float64)
    (labels, lw, bt, bw, npos, nneg, lvl_anchor) = captured['t']
    out['num_total_pos'] = npos
    out['num_total_neg'] = nneg
    for l in range(5):
        out['labels_%d' % l] = labels[l].numpy()
        out['label_weights_%d' % l] = lw[l].numpy()
        out['bbox_targets_%d' % l] = bt[l].numpy()
        out['bbox_weights_%d' % l] = bw[l].numpy()
        for b in range(B):
            out['valid_%d_%d' % (b, l)] = captured['valid'][b][l].numpy()
        for nm, tl in (('cls', tc), ('reg', tr), ('iou', ti)):
            g = tl[l].grad.numpy().reshape(-1)
            key = 'g_%s_%d' % (nm, l)
            out[key + '_idx'] = rs.choice(g.size, min(g.size, 3000), replace=False).astype(np.int64)
            out[key] = g[out[key + '_idx']]
            out[key + '_sum'] = np.float64(g.astype(np.float64).sum())
            out[key + '_abs'] = np.float64(np.abs(g.astype(np.float64)).sum())
    inval = [int((captured['valid'][1][l] == 0).sum()) for l in range(5)]
    print('mixed pad: losses', {k: out[k] for k in ('loss_cls', 'loss_bbox', 'losses_iou')},
          'num_total_pos', npos, 'invalid anchors of image 1 per level', inval)
    assert sum(inval) > 0 and all(int((captured['valid'][0][l] == 0).sum()) == 0 for l in range(5))
    save('losses_mixed_pad', **out)



# ---------------------------------------------------------------- IoU-balanced losses (8f.4)
def gen_losses_balanced():
    """head.loss with loss_cls = IOUbalancedSigmoidFocalLoss(eta=1.5) and loss_bbox =
    IoUbalancedSmoothL1Loss(beta=0.11, delta=1.5, loss_weight=3.049) -- the values the target
    configs keep in comments -- on the inputs of losses_small (same targets)."""
    kw = dict(HEAD_KW)
    kw['loss_cls'] = dict(type='IOUbalancedSigmoidFocalLoss', use_sigmoid=True, gamma=2.0,
                          alpha=0.25, eta=1.5, loss_weight=1.0)
    kw['loss_bbox'] = dict(type='IoUbalancedSmoothL1Loss', beta=0.11, delta=1.5, loss_weight=3.049)
    head = IoUawareRetinaHead(**kw)
    assert head.IoU_balanced_Cls and head.IoU_balanced_Loc
    train_cfg = ref_shim.to_cfg(dict(
        assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0,
                      ignore_iof_thr=-1),
        allowed_border=-1, pos_weight=-1, debug=False))
    seed, B, ih, iw, ph, pw = 303, 2, 120, 157, 128, 160
    cls, reg, iou = synth.head_outputs(seed, B, ph, pw, 'A')
    gts, gls = synth.train_targets(seed + 1, B, ih, iw)
    metas = [synth.img_meta(ih, iw, ph, pw, 1.0) for _ in range(B)]
    out = dict(seed=seed, batch=B, img=np.array([ih, iw, ph, pw]), kind='A',
               checksum=synth.checksum(cls + reg + iou), eta=np.float32(1.5),
               delta=np.float32(1.5), beta=np.float32(0.11), bbox_loss_weight=np.float32(3.049))
    tc = [torch.from_numpy(x).requires_grad_(True) for x in cls]
    tr = [torch.from_numpy(x).requires_grad_(True) for x in reg]
    ti = [torch.from_numpy(x).requires_grad_(True) for x in iou]
    losses = head.loss(tc, tr, ti, [torch.from_numpy(g) for g in gts],
                       [torch.from_numpy(g) for g in gls], metas, train_cfg)
    sum(sum(v) for v in losses.values()).backward()
    for k, v in losses.items():
        out[k] = np.array([float(x) for x in v], np.float64)
    rs = np.random.RandomState(98)
    for l in range(5):
        for nm, tl in (('cls', tc), ('reg', tr), ('iou', ti)):
            g = tl[l].grad.numpy().reshape(-1)
            key = 'g_%s_%d' % (nm, l)
            out[key + '_idx'] = rs.choice(g.size, min(g.size, 3000), replace=False).astype(np.int64)
            out[key] = g[out[key + '_idx']]
            out[key + '_sum'] = np.float64(g.astype(np.float64).sum())
            out[key + '_abs'] = np.float64(np.abs(g.astype(np.float64)).sum())
    print('balanced losses', {k: out[k] for k in ('loss_cls', 'loss_bbox', 'losses_iou')})
    save('losses_balanced', **out)


# ---------------------------------------------------------------- end to end (config 1; I12, I13, f.1)
E2E_CASES = (
    # name, weight seed, image seed, pad h, pad w, img h, img w, scale_factor
    ('small', 77, 5, 256, 320, 250, 317, 1.25),
    ('full', 77, 6, 800, 1344, 800, 1333, 1.0),
)
COCO_CAT_IDS = [i for i in range(1, 91) if i not in (12, 26, 29, 30, 45, 66, 68, 69, 71, 83)]


class _FakeCoco(object):
    """what det2json reads from a CocoDataset (coco_utils.py:103-113): len, img_ids, cat_ids"""

    def __init__(self, img_ids):
        self.img_ids, self.cat_ids = list(img_ids), list(COCO_CAT_IDS)

    def __len__(self):
        return len(self.img_ids)


def gen_e2e():
    """the reference's own test-time call (tools/test.py:25 -> base.py:62-123 ->
    single_stage.py:64-96) on a whole R-50 detector with the deterministic 'trained-like'
    weights of synth.e2e_fill_state: image -> per-class result arrays."""
    import json
    from mmdet.models import build_detector
    from mmdet.core.evaluation import coco_utils
    rcfg = ref_shim.load_config(ref_shim.REF + '/configs/iou_aware_single_stage_detector/'
                                'iou_aware_retinanet_r50_fpn_1x_4gpu.py')
    rcfg.model['pretrained'] = None
    assert len(COCO_CAT_IDS) == 80
    for name, wseed, iseed, ph, pw, ih, iw, sf in E2E_CASES:
        torch.manual_seed(0)
        ref = build_detector(rcfg.model, train_cfg=rcfg.train_cfg, test_cfg=rcfg.test_cfg).eval()
        with torch.no_grad():
            synth.e2e_fill_state(ref.state_dict(), wseed)
        img = synth.e2e_image(iseed, 1, ph, pw, ih, iw)
        ori = (int(round(ih / sf)), int(round(iw / sf)), 3)
        meta = dict(ori_shape=ori, img_shape=(ih, iw, 3), pad_shape=(ph, pw, 3), scale_factor=sf,
                    flip=False)
        g, l = synth.e2e_gts(iseed + 100, ih, iw)
        x = torch.from_numpy(img)
        # head outputs, for the conv-level tolerance check on the GPU
        with torch.no_grad():
            cls, reg, iou = ref.bbox_head(ref.extract_feat(x))
        out = dict(weight_seed=wseed, image_seed=iseed, img=np.array([ih, iw, ph, pw]),
                   scale_factor=np.float32(sf), ori_shape=np.array(ori),
                   gt_bboxes=g, gt_labels=l, img_checksum=synth.checksum([img]),
                   weight_checksum=synth.checksum(
                       [v.numpy() for k, v in sorted(ref.state_dict().items())]))
        rs = np.random.RandomState(1000 + iseed)
        for nm, ts in (('cls', cls), ('reg', reg), ('iou', iou)):
            for lv, t in enumerate(ts):
                a = t.numpy().reshape(-1)
                idx = rs.choice(a.size, min(a.size, 2048), replace=False).astype(np.int64)
                out['%s_idx_%d' % (nm, lv)] = idx
                out['%s_val_%d' % (nm, lv)] = a[idx]
                out['%s_rms_%d' % (nm, lv)] = np.float64(np.sqrt((a.astype(np.float64) ** 2).mean()))
        import mmdet.models.detectors.single_stage as ref_ss
        b2r, b2r_in = ref_ss.bbox2result, []
        ref_ss.bbox2result = lambda d, lb, n: (b2r_in.append((d.clone(), lb.clone())), b2r(d, lb, n))[1]
        with Capture() as cap, torch.no_grad():
            result = ref(return_loss=False, rescale=True, img=[x], img_meta=[[meta]],
                         gt_bboxes=[[torch.from_numpy(g)]], gt_labels=[[torch.from_numpy(l)]])
        ref_ss.bbox2result = b2r
        out['det_bboxes'], out['det_labels'] = b2r_in[0][0].numpy(), b2r_in[0][1].numpy()
        assert isinstance(result, list) and len(result) == 80
        mb, ms = cap.mlvl[0]
        # candidate indices per level, in the reference's order
        topk_inds, ti, margins = [], 0, []
        for lv in range(5):
            Nl = cls[lv].shape[2] * cls[lv].shape[3] * synth.A
            if Nl > 1000:
                ms_l, idx = cap.topk[ti]
                ti += 1
                topk_inds.append(idx.numpy().astype(np.int32))
                srt = ms_l.sort(descending=True).values[:1001].double()
                margins.append(float(((srt[:-1] - srt[1:]) / srt[:-1]).min()))
                out['topk_cut_margin_%d' % lv] = np.float64((srt[999] - srt[1000]) / srt[999])
            else:
                topk_inds.append(np.arange(Nl, dtype=np.int32))
        out['topk_inds'] = np.concatenate(topk_inds)
        out['topk_margin'] = np.array(margins)
        # all survivors of the 80 NMS problems, ranked like multiclass_nms ranks them (bbox_nms.py:52-56)
        surv = []
        ci = 0
        for c in range(80):
            mask = ms[:, c + 1] > 0.05
            if not mask.any():
                continue
            dets_c, inds = cap.nms_calls[ci]
            ci += 1
            rows = torch.nonzero(mask).squeeze(1)[inds]
            surv += [(float(dets_c[i, 4]), c, int(r)) for i, r in zip(inds.tolist(), rows.tolist())]
        surv.sort(key=lambda t: -t[0])
        out['num_survivors'] = len(surv)
        out['into_nms'] = int((ms[:, 1:] > 0.05).sum())
        sc = np.array([s[0] for s in surv[:101]], np.float64)
        out['det_score_gaps'] = (sc[:-1] - sc[1:]) / sc[:-1]          # relative gaps of the ranking
        out['det_rows'] = np.array([s[2] for s in surv[:100]], np.int32)
        out['det_classes'] = np.array([s[1] for s in surv[:100]], np.int32)
        out['result_counts'] = np.array([r.shape[0] for r in result], np.int32)
        out['result_cat'] = np.concatenate(result, 0).astype(np.float32)
        assert all(r.dtype == np.float32 and r.shape[1] == 5 for r in result)
        js = coco_utils.det2json(_FakeCoco([397133]), [result])
        out['coco_json'] = np.array(json.dumps(js))
        out['coco_cat_ids'] = np.array(COCO_CAT_IDS)
        print('e2e', name, 'dets', int(out['result_counts'].sum()), 'survivors', len(surv),
              'into-nms', out['into_nms'], 'topk margins', margins,
              'min det gap %.2e' % out['det_score_gaps'].min(),
              'cut gap %.2e' % out['det_score_gaps'][-1])
        save('e2e_' + name, **out)


# ---------------------------------------------------------------- deeper backbones (configs 3, 4)
E2E_BACKBONES = (
    # name, reference config file, backbone overrides, (image seed, pad h, pad w, img h, img w)
    ('r101', 'iou_aware_retinanet_r101_fpn_1x_4gpu.py', {}, (9, 256, 320, 250, 317)),
    ('x101_32x4d', 'iou_aware_retinanet_x101_32x4d_fpn_1x_4gpu.py', {}, (9, 256, 320, 250, 317)),
    ('x101_64x4d', 'iou_aware_retinanet_x101_32x4d_fpn_1x_4gpu.py', dict(groups=64, base_width=4),
     (9, 256, 320, 250, 317)),
    # BASELINE configs 3 / 4 at the benchmark's size
    ('r101_full', 'iou_aware_retinanet_r101_fpn_1x_4gpu.py', {}, (12, 800, 1344, 800, 1333)),
    ('x101_64x4d_full', 'iou_aware_retinanet_x101_32x4d_fpn_1x_4gpu.py',
     dict(groups=64, base_width=4), (13, 800, 1344, 800, 1333)),
)


def gen_e2e_backbones(only=None):
    """R-101 (BASELINE config 3's backbone) and ResNeXt-101 32x4d / 64x4d (config 4: the 32x4d
    config file of the reference with groups=64, as retinanet_x101_64x4d_fpn_1x.py sets them):
    the reference detector on the trained-like weights, one image (256x320, and 800x1344 for the
    two BASELINE backbones) through its test-time call -- sampled head logits and the per-class
    result arrays, TWICE: as the reference computes them (fp32, oneDNN on this CPU) and with the
    same reference modules converted to fp64 (`*64` keys: the "true" convolution values, which
    say whose fp32 rounding a deviation belongs to)."""
    from mmdet.models import build_detector
    for name, cfile, over, (iseed, ph, pw, ih, iw) in E2E_BACKBONES:
        if only and name not in only:
            continue
        rcfg = ref_shim.load_config(ref_shim.REF + '/configs/iou_aware_single_stage_detector/' + cfile)
        rcfg.model['pretrained'] = None
        rcfg.model['backbone'].update(over)
        torch.manual_seed(0)
        ref = build_detector(rcfg.model, train_cfg=rcfg.train_cfg, test_cfg=rcfg.test_cfg).eval()
        with torch.no_grad():
            synth.e2e_fill_state(ref.state_dict(), 77)
        sf = 1.0
        img = synth.e2e_image(iseed, 1, ph, pw, ih, iw)
        meta = dict(ori_shape=(ih, iw, 3), img_shape=(ih, iw, 3), pad_shape=(ph, pw, 3),
                    scale_factor=sf, flip=False)
        g, l = synth.e2e_gts(iseed + 100, ih, iw)
        x = torch.from_numpy(img)
        with torch.no_grad():
            cls, reg, iou = ref.bbox_head(ref.extract_feat(x))
            result = ref(return_loss=False, rescale=True, img=[x], img_meta=[[meta]],
                         gt_bboxes=[[torch.from_numpy(g)]], gt_labels=[[torch.from_numpy(l)]])
        out = dict(weight_seed=77, image_seed=iseed, img=np.array([ih, iw, ph, pw]),
                   scale_factor=np.float32(sf), ori_shape=np.array([ih, iw, 3]), gt_bboxes=g,
                   gt_labels=l, img_checksum=synth.checksum([img]),
                   weight_checksum=synth.checksum(
                       [v.numpy() for k, v in sorted(ref.state_dict().items())]),
                   backbone=np.array(str(dict(rcfg.model['backbone']))))
        rs = np.random.RandomState(2000 + iseed)
        for nm, ts in (('cls', cls), ('reg', reg), ('iou', iou)):
            for lv, t in enumerate(ts):
                a = t.numpy().reshape(-1)
                idx = rs.choice(a.size, min(a.size, 1024), replace=False).astype(np.int64)
                out['%s_idx_%d' % (nm, lv)] = idx
                out['%s_val_%d' % (nm, lv)] = a[idx]
        out['result_counts'] = np.array([r.shape[0] for r in result], np.int32)
        out['result_cat'] = np.concatenate(result, 0).astype(np.float32)
        # the same reference modules in fp64: the fp32 weights / image converted exactly
        ref64 = ref.double()
        x64 = x.double()
        with torch.no_grad():
            cls64, reg64, iou64 = ref64.bbox_head(ref64.extract_feat(x64))
            result64 = ref64(return_loss=False, rescale=True, img=[x64], img_meta=[[meta]],
                             gt_bboxes=[[torch.from_numpy(g).double()]],
                             gt_labels=[[torch.from_numpy(l)]])
        assert cls64[0].dtype == torch.float64 and result64[0].dtype == np.float64
        ref_err = 0.0
        for nm, ts in (('cls', cls64), ('reg', reg64), ('iou', iou64)):
            for lv, t in enumerate(ts):
                v = t.numpy().reshape(-1)[out['%s_idx_%d' % (nm, lv)]]
                out['%s_val64_%d' % (nm, lv)] = v
                e = np.abs(out['%s_val_%d' % (nm, lv)].astype(np.float64) - v) / np.maximum(1.0, np.abs(v))
                ref_err = max(ref_err, float(e.max()))
        out['result_counts64'] = np.array([r.shape[0] for r in result64], np.int32)
        out['result_cat64'] = np.concatenate(result64, 0).astype(np.float64)
        out['ref_logit_err_vs_fp64'] = np.float64(ref_err)
        sc = np.sort(out['result_cat'][:, 4].astype(np.float64))[::-1]
        print('e2e backbone', name, 'dets', int(out['result_counts'].sum()),
              'fp64 dets', int(out['result_counts64'].sum()),
              'min relative score gap %.2e' % float(((sc[:-1] - sc[1:]) / sc[:-1]).min()),
              'reference fp32 vs its fp64 evaluation, sampled head logits: %.2e' % ref_err)
        save('e2e_backbone_' + name, **out)


# ---------------------------------------------------------------- T4 pin
def gen_focal_op():
    """SigmoidFocalLoss op (T4): the CUDA kernel has no CPU twin in the reference, but for integer
    targets and unit weights py_sigmoid_focal_loss (losses.py:226-247) -- the function the
    IoU-aware head really uses -- computes the same quantity on a one-hot target.  Stored: the
    elementwise loss and its autograd gradient under a random upstream gradient."""
    rs = np.random.RandomState(41)
    N, C = 131, 80
    x = (rs.standard_normal((N, C)) * 3.0 - 2.0).astype(np.float32)
    x[0, :8] = [-30, -12, -1e-3, 0.0, 1e-3, 12, 30, 80]           # extremes
    t = rs.randint(0, C + 1, N).astype(np.int64)                   # 0 = background
    t[:8] = [1, 2, 3, 4, 5, 6, 7, 8]
    up = rs.uniform(0.1, 2.0, (N, C)).astype(np.float32)
    out = dict(seed=41, logits=x, targets=t, upstream=up)
    for gamma, alpha in ((2.0, 0.25), (1.5, 0.4), (0.0, 0.5)):
        tx = torch.from_numpy(x).clone().requires_grad_(True)
        onehot = torch.zeros(N, C)
        pos = torch.nonzero(torch.from_numpy(t) >= 1).squeeze(1)
        onehot[pos, torch.from_numpy(t)[pos] - 1] = 1
        loss = ref_losses.py_sigmoid_focal_loss(tx, onehot, torch.ones(N, C), gamma=gamma,
                                                alpha=alpha, reduction='none')
        (loss * torch.from_numpy(up)).sum().backward()
        tag = 'g%g_a%g' % (gamma, alpha)
        out['loss_' + tag] = loss.detach().numpy()
        out['grad_' + tag] = tx.grad.numpy()
        # the same function evaluated in the OTHER storage types of the reference op's dispatch
        # (sigmoid_focal_loss_cuda.cu:115,151 AT_DISPATCH_FLOATING_TYPES_AND_HALF: intermediates in
        # scalar_t): double, and half on the half-rounded logits -- ADVICE r4: the dtype-generic HIP op
        # computes in fp32 and rounds once; these pin how far that is from scalar_t arithmetic
        for name, dt in (('64', torch.float64), ('16', torch.float16)):
            td = torch.from_numpy(x).to(dt).clone().requires_grad_(True)
            ld = ref_losses.py_sigmoid_focal_loss(td, onehot.to(dt), torch.ones(N, C, dtype=dt), gamma=gamma,
                                                  alpha=alpha, reduction='none')
            (ld * torch.from_numpy(up).to(dt)).sum().backward()
            out['loss%s_%s' % (name, tag)] = ld.detach().numpy()
            out['grad%s_%s' % (name, tag)] = td.grad.numpy()
    out['params'] = np.array([[2.0, 0.25], [1.5, 0.4], [0.0, 0.5]], np.float32)
    save('focal_op', **out)


# ---------------------------------------------------------------- training iteration (config 5)
def gen_train_e2e():
    """one training iteration of the REFERENCE detector (tools/train.py -> batch_processor ->
    model(**data) with return_loss=True, mmdet/apis/train.py:38-45; ResNet.train() with
    norm_eval=True / frozen_stages=1, resnet.py:520-527) on the deterministic trained-like
    weights: the loss dict, the gradient norm of EVERY trainable parameter and sampled gradient
    entries of a few of them -- what the fused training route must reproduce."""
    from mmdet.models import build_detector
    rcfg = ref_shim.load_config(ref_shim.REF + '/configs/iou_aware_single_stage_detector/'
                                'iou_aware_retinanet_r50_fpn_1x_4gpu.py')
    rcfg.model['pretrained'] = None
    wseed, iseed, B, ph, pw, ih, iw = 7, 3, 2, 256, 320, 250, 317
    torch.manual_seed(0)
    ref = build_detector(rcfg.model, train_cfg=rcfg.train_cfg, test_cfg=rcfg.test_cfg)
    with torch.no_grad():
        synth.e2e_fill_state(ref.state_dict(), wseed)
    ref.train()
    img = synth.e2e_image(iseed, B, ph, pw, ih, iw)
    gts, gls = synth.train_targets(11, B, ih, iw, max_gt=6)
    metas = [synth.img_meta(ih, iw, ph, pw, 1.0) for _ in range(B)]
    losses = ref(torch.from_numpy(img), metas, return_loss=True,
                 gt_bboxes=[torch.from_numpy(g) for g in gts],
                 gt_labels=[torch.from_numpy(g) for g in gls])
    total = sum(sum(v) for k, v in losses.items() if 'loss' in k)
    total.backward()
    out = dict(weight_seed=wseed, image_seed=iseed, target_seed=11, batch=B,
               img=np.array([ih, iw, ph, pw]), img_checksum=synth.checksum([img]),
               total=np.float64(float(total)))
    for k, v in losses.items():
        out[k] = np.array([float(x) for x in v], np.float64)
    names, norms = [], []
    rs = np.random.RandomState(5)
    sampled = ('backbone.layer2.0.conv1.weight', 'backbone.layer2.0.bn1.weight',
               'backbone.layer2.0.bn1.bias', 'backbone.layer2.0.conv2.weight',
               'backbone.layer2.0.downsample.0.weight', 'backbone.layer2.3.conv3.weight',
               'backbone.layer3.2.conv2.weight', 'backbone.layer3.5.bn3.weight',
               'backbone.layer4.0.conv2.weight', 'backbone.layer4.2.conv1.weight',
               'neck.lateral_convs.0.conv.weight', 'neck.lateral_convs.2.conv.bias',
               'neck.fpn_convs.1.conv.weight', 'neck.fpn_convs.3.conv.weight',
               'bbox_head.cls_convs.0.conv.weight', 'bbox_head.reg_convs.3.conv.bias',
               'bbox_head.retina_cls.weight', 'bbox_head.retina_reg.weight',
               'bbox_head.retina_iou.weight', 'bbox_head.retina_cls.bias')
    for name, p in ref.named_parameters():
        if p.grad is None:
            assert not p.requires_grad, name
            continue
        g = p.grad.numpy().astype(np.float64).reshape(-1)
        names.append(name)
        norms.append(float(np.sqrt((g * g).sum())))
        if name in sampled:
            idx = rs.choice(g.size, min(g.size, 2000), replace=False).astype(np.int64)
            out['gidx/' + name] = idx
            out['gval/' + name] = g[idx].astype(np.float32)
    assert all(n in names for n in sampled), [n for n in sampled if n not in names]
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms, np.float64)
    out['frozen'] = np.array([n for n, p in ref.named_parameters() if not p.requires_grad])
    print('train_e2e: total %.6f, %d trainable / %d frozen parameters' % (
        float(total), len(names), len(out['frozen'])), {k: out[k] for k in losses})
    save('train_e2e', **out)


# ---------------------------------------------------------------- model structure (B1, I3)
def gen_model():
    """parameter names / shapes of the four reference configs (+ the 64x4d backbone of BASELINE
    config 4) and a tiny forward check value, so the GPU box can check checkpoint compatibility."""
    import glob
    import json
    from mmdet.models import build_detector
    out = {}
    files = sorted(glob.glob(ref_shim.REF + '/configs/iou_aware_single_stage_detector/*.py'))
    for f in files:
        cfg = ref_shim.load_config(f)
        cfg.model['pretrained'] = None
        torch.manual_seed(0)
        m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
        name = os.path.basename(f)[:-3]
        out[name] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    with open(os.path.join(HERE, 'state_dict_keys.json'), 'w') as fh:
        json.dump(out, fh)
    print('wrote state_dict_keys.json', {k: len(v) for k, v in out.items()})


if __name__ == '__main__':
    which = sys.argv[1:] or ['anchors', 'delta2bbox', 'nms', 'get_bboxes', 'soft_nms', 'losses', 'losses_balanced', 'model', 'e2e', 'focal_op', 'train_e2e', 'e2e_backbones', 'mnms_quirk', 'get_bboxes_vecscale', 'nms_f64', 'mnms_big',
                             'losses_mixed_pad', 'get_bboxes_softmax']
    for w in which:
        if ':' in w:                          # e.g. e2e_backbones:r101_full,x101_64x4d_full
            w, only = w.split(':')
            globals()['gen_' + w](only.split(','))
        else:
            globals()['gen_' + w]()
