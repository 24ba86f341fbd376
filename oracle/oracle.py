"""TEST INFRASTRUCTURE ONLY.  ctypes front-end of the CPU oracle.

May be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libiouaware_oracle.so')
_lib = None

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in
            ('iouaware_oracle.c', 'iouaware_oracle_loss.c', 'iouaware_oracle_softnms.c', 'iouaware_oracle_preproc.c',
             'ia_oracle_math.h', 'Makefile')]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.run(['make', '-C', _HERE, '-B', '_build/libiouaware_oracle.so'], check=True,
                   stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.ia_o_count_sigmoid_nonmonotone.restype = C.c_longlong
        _lib.ia_o_count_sigmoid_nonmonotone.argtypes = [C.c_float, C.c_float]
        _lib.ia_o_focal_loss.restype = C.c_double
        _lib.ia_o_smooth_l1.restype = C.c_double
        _lib.ia_o_iou_bce.restype = C.c_double
        _lib.ia_o_focal_loss_op_fwd.restype = None
        _lib.ia_o_focal_loss_op_bwd.restype = None
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _ip(a):
    return a.ctypes.data_as(_i32p)


def vec(fn, x):
    x = _f(x)
    y = np.empty_like(x)
    getattr(lib(), 'ia_o_vec_' + fn)(_fp(x), _fp(y), C.c_longlong(x.size))
    return y


def gen_base_anchors(base_size, scales, ratios):
    scales = _f(scales)
    ratios = _f(ratios)
    out = np.empty((len(ratios) * len(scales), 4), np.float32)
    lib().ia_o_gen_base_anchors(C.c_float(base_size), _fp(scales), len(scales), _fp(ratios),
                                len(ratios), _fp(out))
    return out


def grid_anchors(base, H, W, stride):
    base = _f(base)
    out = np.empty((H * W * base.shape[0], 4), np.float32)
    lib().ia_o_grid_anchors(_fp(base), base.shape[0], H, W, stride, _fp(out))
    return out


def delta2bbox(rois, deltas, means=(0, 0, 0, 0), stds=(1, 1, 1, 1), max_shape=None):
    rois, deltas = _f(rois), _f(deltas)
    means, stds = _f(means), _f(stds)
    out = np.empty_like(deltas)
    has = 0 if max_shape is None else 1
    h, w = (0, 0) if max_shape is None else (max_shape[0], max_shape[1])
    lib().ia_o_delta2bbox(_fp(rois), _fp(deltas), rois.shape[0], _fp(means), _fp(stds), has,
                          C.c_float(h), C.c_float(w), _fp(out))
    return out


def nms(dets, thr):
    dets = _f(dets).reshape(-1, 5)
    keep = np.empty(max(dets.shape[0], 1), np.int32)
    m = lib().ia_o_nms(_fp(dets), dets.shape[0], C.c_float(thr), _ip(keep))
    return keep[:m].astype(np.int64)


def head_base_anchors(strides, octave_base_scale=4, scales_per_octave=3,
                      ratios=(0.5, 1.0, 2.0)):
    """anchor_scales of IoUawareRetinaHead.__init__ (iou_aware_retina_head.py:81-83)."""
    octave = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
    scales = (octave * octave_base_scale).astype(np.float32)   # torch.Tensor(scales) -> fp32
    return np.stack([gen_base_anchors(s, scales, ratios) for s in strides])


def get_bboxes_single(cls, reg, iou, strides, base_anchors, img_shape, scale_factor, rescale,
                      nms_pre=1000, score_thr=0.05, iou_thr=0.5, max_per_img=100,
                      means=(0, 0, 0, 0), stds=(1, 1, 1, 1), C_cls=80, softmax=False):
    """cls/reg/iou: lists (one per level) of (ch,H,W) fp32 arrays of ONE image.
    softmax=True: use_sigmoid_cls=False (iou_aware_retina_head.py:506-507,540-541): cls has
    A * (C_cls + 1) channels, channel 0 of an anchor = background; C_cls foreground columns."""
    L = len(cls)
    cls = [_f(x) for x in cls]
    reg = [_f(x) for x in reg]
    iou = [_f(x) for x in iou]
    A = iou[0].shape[0]
    Hs = np.array([x.shape[1] for x in iou], np.int32)
    Ws = np.array([x.shape[2] for x in iou], np.int32)
    st = np.array(strides, np.int32)
    Nl = [int(h) * int(w) * A for h, w in zip(Hs, Ws)]
    kl = [min(nms_pre, n) if nms_pre > 0 else n for n in Nl]
    N, R = sum(Nl), sum(kl)
    base = _f(base_anchors)
    sf = np.asarray(scale_factor, np.float32).reshape(-1)
    sf = _f(np.repeat(sf, 4) if sf.size == 1 else sf)
    means, stds = _f(means), _f(stds)
    PP = _f32p * L
    rowmax = np.empty(N, np.float32)
    topk = np.empty(R, np.int32)
    boxes = np.empty((R, 4), np.float32)
    scores = np.empty((R, C_cls), np.float32)
    kc = np.zeros(C_cls, np.int32)
    kr = np.zeros((C_cls, R), np.int32)
    mp = max(max_per_img, 0) if max_per_img >= 0 else R * C_cls
    db = np.zeros((max(mp, 1), 5), np.float32)
    dl = np.zeros(max(mp, 1), np.int32)
    dr = np.zeros(max(mp, 1), np.int32)
    Rout = C.c_int32(0)
    assert all(x.shape[0] == A * (C_cls + (1 if softmax else 0)) for x in cls)
    fn = lib().ia_o_get_bboxes_single_ex
    fn.restype = C.c_int
    nd = fn(L, PP(*[_fp(x) for x in cls]), PP(*[_fp(x) for x in reg]),
            PP(*[_fp(x) for x in iou]), _ip(Hs), _ip(Ws), _ip(st), _fp(base), A, C_cls,
            _fp(means), _fp(stds), C.c_float(img_shape[0]), C.c_float(img_shape[1]), _fp(sf),
            int(bool(rescale)), int(nms_pre), C.c_float(score_thr), C.c_float(iou_thr),
            int(max_per_img), _fp(rowmax), _ip(topk), _fp(boxes), _fp(scores), _ip(kc), _ip(kr),
            _fp(db), _ip(dl), _ip(dr), C.byref(Rout), int(bool(softmax)))
    assert Rout.value == R
    lvl_off = np.cumsum([0] + Nl)
    cand_off = np.cumsum([0] + kl)
    return dict(rowmax=rowmax, topk_inds=topk, mlvl_bboxes=boxes, mlvl_scores=scores,
                keep_count=kc, keep_rows=kr, det_bboxes=db[:nd].copy(),
                det_labels=dl[:nd].astype(np.int64), det_rows=dr[:nd].copy(), num_det=nd,
                level_off=lvl_off, cand_off=cand_off)


def sigmoid_nonmonotone_count(lo, hi):
    return int(lib().ia_o_count_sigmoid_nonmonotone(C.c_float(lo), C.c_float(hi)))


# ------------------------------------------------------------------ training losses
def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def focal_loss(cls, labels, label_weights, A, gamma=2.0, alpha=0.25, gscale=None):
    """cls (B, A*C, H, W); labels (B*N_l) int64; label_weights (B*N_l).
    -> (sum, grad or None); grad = gscale * d sum / d cls."""
    cls = _f(cls)
    B, ch, H, W = cls.shape
    Cn = ch // A
    labels = np.ascontiguousarray(labels, np.int64).reshape(-1)
    lw = _f(label_weights).reshape(-1)
    grad = np.empty_like(cls) if gscale is not None else None
    s = lib().ia_o_focal_loss(_fp(cls), _i64p(labels), _fp(lw), B, A, Cn, H * W, C.c_float(gamma),
                              C.c_float(alpha), C.c_float(np.float32(1.0 - alpha)),
                              C.c_float(0.0 if gscale is None else gscale),
                              _fp(grad) if grad is not None else None)
    return s, grad


def smooth_l1(pred, target, weight, A, beta, gscale=None):
    pred = _f(pred)
    B, ch, H, W = pred.shape
    target, weight = _f(target), _f(weight)
    grad = np.empty_like(pred) if gscale is not None else None
    s = lib().ia_o_smooth_l1(_fp(pred), _fp(target), _fp(weight), B, A, H * W, C.c_float(beta),
                             C.c_float(0.0 if gscale is None else gscale),
                             _fp(grad) if grad is not None else None)
    return s, grad


def iou_bce(bbox_pred, iou_pred, bbox_targets, bbox_weights, base, stride, means=(0, 0, 0, 0),
            stds=(1, 1, 1, 1), gscale=None, attach=True):
    """-> (sum, iou_target (B*N_l), grad_iou_pred, grad_bbox_pred)"""
    bbox_pred, iou_pred = _f(bbox_pred), _f(iou_pred)
    B, A, H, W = iou_pred.shape
    bt, bw, base = _f(bbox_targets), _f(bbox_weights), _f(base)
    means, stds = _f(means), _f(stds)
    tgt = np.empty(B * A * H * W, np.float32)
    g_iou = np.empty_like(iou_pred) if gscale is not None else None
    g_box = np.empty_like(bbox_pred) if (gscale is not None and attach) else None
    s = lib().ia_o_iou_bce(_fp(bbox_pred), _fp(iou_pred), _fp(bt), _fp(bw), _fp(base), B, A, H, W,
                           int(stride), _fp(means), _fp(stds),
                           C.c_float(0.0 if gscale is None else gscale), _fp(tgt),
                           _fp(g_iou) if g_iou is not None else None,
                           _fp(g_box) if g_box is not None else None)
    return s, tgt, g_iou, g_box


def focal_loss_op(logits, targets, gamma, alpha, d_losses=None):
    logits = _f(logits)
    N, Cn = logits.shape
    targets = np.ascontiguousarray(targets, np.int64)
    out = np.empty_like(logits)
    if d_losses is None:
        lib().ia_o_focal_loss_op_fwd(_fp(logits), _i64p(targets), N, Cn, C.c_float(gamma),
                                     C.c_float(alpha), _fp(out))
    else:
        d = _f(d_losses)
        lib().ia_o_focal_loss_op_bwd(_fp(logits), _i64p(targets), _fp(d), N, Cn, C.c_float(gamma),
                                     C.c_float(alpha), _fp(out))
    return out


# ------------------------------------------------------------------ soft-NMS
SOFT_METHODS = {'linear': 1, 'gaussian': 2}


def soft_nms(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    """soft_nms_cpu.pyx:22-127 -> (new_dets (m,5) fp32 with decayed scores, inds (m,) int64)."""
    dets = _f(dets).reshape(-1, 5)
    n = dets.shape[0]
    out = np.zeros((max(n, 1), 5), np.float32)
    inds = np.zeros(max(n, 1), np.int32)
    m = lib().ia_o_soft_nms(_fp(dets), C.c_int(n), C.c_float(iou_thr),
                            C.c_int(SOFT_METHODS.get(method, method) if isinstance(method, str)
                                    else int(method)),
                            C.c_float(sigma), C.c_float(min_score), _fp(out), _ip(inds))
    return out[:m].copy(), inds[:m].astype(np.int64)


def multiclass_soft_nms(bboxes, scores, score_thr, iou_thr, method='linear', sigma=0.5,
                        min_score=1e-3, max_per_img=100):
    """bbox_nms.py:29-56 with nms.type='soft_nms'; scores (R,C) without background column.
    -> dict(det_bboxes (k,5), det_labels (k,), det_rows (k,), keep_count (C,), keep_rows (C,R),
    keep_scores (C,R))"""
    bboxes = _f(bboxes).reshape(-1, 4)
    scores = _f(scores)
    R, Cn = scores.shape
    kc = np.zeros(Cn, np.int32)
    kr = np.full((Cn, max(R, 1)), -1, np.int32)
    ks = np.zeros((Cn, max(R, 1)), np.float32)
    cap = max_per_img if max_per_img >= 0 else R * Cn
    db = np.zeros((max(cap, 1), 5), np.float32)
    dl = np.zeros(max(cap, 1), np.int32)
    dr = np.zeros(max(cap, 1), np.int32)
    nd = lib().ia_o_multiclass_soft_nms(
        _fp(bboxes), _fp(scores), C.c_int(R), C.c_int(Cn), C.c_float(score_thr),
        C.c_float(iou_thr), C.c_int(SOFT_METHODS[method]), C.c_float(sigma), C.c_float(min_score),
        C.c_int(max_per_img), _ip(kc), _ip(kr), _fp(ks), _fp(db), _ip(dl), _ip(dr))
    return dict(det_bboxes=db[:nd].copy(), det_labels=dl[:nd].astype(np.int64),
                det_rows=dr[:nd].copy(), keep_count=kc, keep_rows=kr, keep_scores=ks)


def vec_exp_f64(x):
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    lib().ia_o_vec_exp_f64(x.ctypes.data_as(dp), y.ctypes.data_as(dp), C.c_longlong(x.size))
    return y


# ------------------------------------------------------------------ IoU-balanced losses (8f.4)
def focal_loss_balanced(cls, labels, label_weights, iou, A, gamma=2.0, alpha=0.25, eta=1.5,
                        gscale=None):
    """iou_balanced_sigmoid_focal_loss (losses.py:309-374) on NCHW logits.
    -> (sum, grad or None, sums3 = [S0, S1, S2])"""
    cls = _f(cls)
    B, ch, H, W = cls.shape
    Cn = ch // A
    labels = np.ascontiguousarray(labels, np.int64).reshape(-1)
    lw, iou = _f(label_weights).reshape(-1), _f(iou).reshape(-1)
    grad = np.empty_like(cls) if gscale is not None else None
    sums = np.zeros(3, np.float64)
    fn = lib().ia_o_focal_loss_balanced
    fn.restype = C.c_double
    s = fn(_fp(cls), _i64p(labels), _fp(lw), _fp(iou), B, A, Cn, H * W, C.c_float(gamma),
           C.c_float(alpha), C.c_float(np.float32(1.0 - alpha)), C.c_float(eta),
           C.c_float(0.0 if gscale is None else gscale), _fp(grad) if grad is not None else None,
           sums.ctypes.data_as(C.POINTER(C.c_double)))
    return s, grad, sums


def smooth_l1_balanced(pred, target, weight, iou, A, beta, delta, gscale=None):
    """weighted_iou_balanced_smoothl1 (losses.py:416-458) -> (sum, grad or None)"""
    pred = _f(pred)
    B, ch, H, W = pred.shape
    target, weight, iou = _f(target), _f(weight), _f(iou).reshape(-1)
    grad = np.empty_like(pred) if gscale is not None else None
    fn = lib().ia_o_smooth_l1_balanced
    fn.restype = C.c_double
    s = fn(_fp(pred), _fp(target), _fp(weight), _fp(iou), B, A, H * W, C.c_float(beta),
           C.c_float(delta), C.c_float(0.0 if gscale is None else gscale),
           _fp(grad) if grad is not None else None)
    return s, grad


# ------------------------------------------------------------------ image pre-processing (8f.3)
def rescale_size(h, w, scale, keep_ratio=True):
    """mmcv.imrescale / imresize sizes (mmcv 0.2.x image/transforms/resize.py, restated):
    -> (new_h, new_w, scale_factor) ; scale_factor a python float (keep_ratio) or a
    (w_scale, h_scale, w_scale, h_scale) fp32 array."""
    if keep_ratio:
        max_long, max_short = max(scale), min(scale)
        sf = min(max_long / max(h, w), max_short / min(h, w))
        return int(h * float(sf) + 0.5), int(w * float(sf) + 0.5), sf
    nw, nh = scale
    return nh, nw, np.array([nw / w, nh / h, nw / w, nh / h], np.float32)


def resize_bilinear_u8(img, nh, nw):
    img = np.ascontiguousarray(img, np.uint8)
    h, w, c = img.shape
    assert c == 3
    out = np.empty((nh, nw, 3), np.uint8)
    u8 = C.POINTER(C.c_uint8)
    lib().ia_o_resize_bilinear_u8(img.ctypes.data_as(u8), h, w, out.ctypes.data_as(u8), nh, nw)
    return out


def image_transform(img, scale, flip=False, keep_ratio=True, mean=(0, 0, 0), std=(1, 1, 1),
                    to_rgb=True, size_divisor=None):
    """ImageTransform.__call__ (mmdet/datasets/transforms.py:31-50)
    -> (img (3,ph,pw) fp32, img_shape, pad_shape, scale_factor)"""
    h, w = img.shape[:2]
    nh, nw, sf = rescale_size(h, w, scale, keep_ratio)
    if nh < 1 or nw < 1:
        # cv2.resize (behind mmcv.imrescale, transforms.py:35) asserts a non-empty dsize: the reference raises
        raise ValueError('rescaled size %d x %d is empty' % (nh, nw))
    r = resize_bilinear_u8(img, nh, nw)
    if size_divisor is not None:
        ph = int(np.ceil(nh / size_divisor)) * size_divisor
        pw = int(np.ceil(nw / size_divisor)) * size_divisor
    else:
        ph, pw = nh, nw
    out = np.empty((3, ph, pw), np.float32)
    u8 = C.POINTER(C.c_uint8)
    lib().ia_o_normalize_flip_pad_chw(r.ctypes.data_as(u8), nh, nw, _fp(_f(mean)), _fp(_f(std)),
                                      int(bool(to_rgb)), int(bool(flip)), ph, pw, _fp(out))
    return out, (nh, nw, 3), (ph, pw, 3), sf
