"""`mmdet.ops.nms.nms` and `mmdet.core.multiclass_nms` with the reference's
call signatures, on the HIP kernels (reference mmdet/ops/nms/nms_wrapper.py:8-49,
mmdet/core/post_processing/bbox_nms.py:6-67)."""
import numpy as np
import torch

from . import ops


def _stage(t, device_id=None):
    """a CPU tensor on the ROCm device the kernels run on (the reference computes such inputs on
    the host, nms_cpu.cpp; this build has no CPU compute path: it stages them to the device and
    back, and raises when there is no device)"""
    if t.is_cuda:
        return t
    if not torch.cuda.is_available():
        raise ops._lib.IouAwareLibraryError(
            'a CPU input needs a ROCm device to be staged to: there is no CPU compute path in this build')
    return t.to('cuda' if device_id is None else 'cuda:{}'.format(device_id))


def nms(dets, iou_thr, device_id=None):
    """dets: (n,5) Tensor (any device; float32 or float64, nms_cpu.cpp:63) or ndarray (with or
    without device_id).  Returns (dets[inds], inds) in the input's type and on the input's device;
    inds ascending (the CPU reference's order, nms_cpu.cpp:58).  CPU tensors and ndarrays -- which
    the reference hands to its C++ loop, nms_wrapper.py:27-45 -- are staged to the ROCm device,
    computed there and brought back."""
    if isinstance(dets, torch.Tensor):
        is_numpy, dets_th = False, dets
    elif isinstance(dets, np.ndarray):
        is_numpy = True
        dets_th = torch.from_numpy(dets)
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(
            type(dets)))
    src = dets_th.device
    if dets_th.shape[0] == 0:
        inds = dets_th.new_zeros(0, dtype=torch.long)
    else:
        inds = ops.nms_indices(_stage(dets_th, device_id), iou_thr).to(src)
    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds


def soft_nms(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    """`mmdet.ops.nms.soft_nms` (nms_wrapper.py:52-78): -> (new_dets (m,5) with decayed scores
    in selection order, inds (m,) int64), in the input's type.  The reference computes on the
    host through numpy whatever the input is; this build computes on the device the tensor
    lives on (a CPU tensor / ndarray is staged to the current ROCm device and back)."""
    if isinstance(dets, torch.Tensor):
        is_tensor, dets_th = True, dets
    elif isinstance(dets, np.ndarray):
        is_tensor, dets_th = False, torch.from_numpy(np.ascontiguousarray(dets, np.float32))
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(
            type(dets)))
    ops._soft_method(method)                               # ValueError first (nms_wrapper.py:65-66)
    src_device = dets_th.device
    if not dets_th.is_cuda:
        if not torch.cuda.is_available():
            raise ops._lib.IouAwareLibraryError('soft_nms needs a ROCm device: no CPU path')
        dets_th = dets_th.cuda()
    new_dets, inds = ops.soft_nms_dets(dets_th, iou_thr, method, sigma, min_score)
    if is_tensor:
        return new_dets.to(device=src_device, dtype=dets.dtype), inds.to(src_device)
    return new_dets.cpu().numpy().astype(np.float32), inds.cpu().numpy().astype(np.int64)


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1,
                   score_factors=None):
    """multi_bboxes (n,4), multi_scores (n,C+1) with the background column first.
    -> (bboxes (k,5), labels (k,) int64).  One batched launch for all classes."""
    if multi_bboxes.shape[1] != 4:
        raise NotImplementedError('class-specific boxes (n, C*4) are a two-stage feature')
    if score_factors is not None:
        raise NotImplementedError('score_factors is unused on this path')
    cfg = dict(nms_cfg)
    nms_type = cfg.pop('type', 'nms')
    if nms_type not in ('nms', 'soft_nms'):
        raise AttributeError("module 'nms_wrapper' has no attribute '%s'" % nms_type)
    n = multi_bboxes.shape[0]
    if n == 0:
        return multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0,), dtype=torch.long)
    src = multi_bboxes.device
    if not multi_bboxes.is_cuda:                           # CPU inputs: staged to the device and back
        multi_bboxes, multi_scores = _stage(multi_bboxes), _stage(multi_scores)
    cap = ops._lib.IA_MAX_PER_IMG
    drop_last = max_num is None or max_num < 0
    if n > ops._lib.IA_MAX_CANDIDATES or (not drop_last and max_num > cap):
        # beyond the batched kernels' capacities: the reference's own structure, one NMS per class
        # (bbox_nms.py:33-56) on the single-problem entry, which takes any n
        b, l = _multiclass_nms_per_class(multi_bboxes, multi_scores, score_thr, nms_type, cfg, max_num)
        return b.to(src), l.to(src)
    if drop_last:
        # reference quirk (bbox_nms.py:52-56): `shape[0] > -1` is always true, so ALL survivors
        # are sorted by score (descending) and `inds[:-1]` drops the globally lowest one.
        # Emulated while the survivor count fits the library's per-image output buffer: the
        # kernel is asked for `cap` rows, which it returns unsorted (class-major) when fewer
        # survive; the sort and the drop happen below.
        max_num_k = cap
    else:
        max_num_k = int(max_num)
    Cn = multi_scores.shape[1] - 1
    Rs = (n + 63) // 64 * 64
    scores_t = multi_bboxes.new_zeros((1, Cn, Rs), dtype=torch.float32)
    scores_t[0, :, :n] = multi_scores[:, 1:].t().to(torch.float32)
    boxes = multi_bboxes.to(torch.float32).reshape(1, n, 4)
    if nms_type == 'soft_nms':
        kw = dict(cfg)
        iou_thr = kw.pop('iou_thr')
        out = ops.multiclass_soft_nms(boxes, scores_t, n, score_thr, iou_thr, max_num_k, **kw)
    else:
        out = ops.multiclass_nms(boxes, scores_t, n, score_thr, cfg['iou_thr'], max_num_k)
    k = int(out[3][0].item())
    if drop_last:
        if k >= cap:
            # more survivors than the batched kernel's output holds: the per-class route
            b, l = _multiclass_nms_per_class(multi_bboxes, multi_scores, score_thr, nms_type, cfg, max_num)
            return b.to(src), l.to(src)
        dets, labels = out[0][0, :k], out[1][0, :k].to(torch.long)
        # stable: equal scores keep their concatenation order (class ascending, row ascending)
        order = torch.sort(dets[:, 4], descending=True, stable=True)[1][:max(k - 1, 0)]
        return dets[order].to(src), labels[order].to(src)
    return out[0][0, :k].to(src), out[1][0, :k].to(torch.long).to(src)


def _multiclass_nms_per_class(multi_bboxes, multi_scores, score_thr, nms_type, cfg, max_num):
    """multiclass_nms as the reference writes it (bbox_nms.py:33-56): a loop over the classes,
    one single-problem NMS each (ia_nms takes any n; soft-NMS up to IA_MAX_CANDIDATES per class),
    class-major concatenation, then the score sort when more than max_num survive (with
    max_num = -1: always, and the lowest survivor is dropped -- the reference's `inds[:-1]`).
    The unbounded route behind the batched kernels' capacities."""
    num_classes = multi_scores.shape[1]
    max_num = -1 if max_num is None else int(max_num)       # like the batched route
    bboxes, labels = [], []
    for i in range(1, num_classes):
        cls_inds = multi_scores[:, i] > score_thr
        if not bool(cls_inds.any()):
            continue
        cls_dets = torch.cat([multi_bboxes[cls_inds, :], multi_scores[cls_inds, i, None]], dim=1)
        if nms_type == 'soft_nms':
            cls_dets, _ = soft_nms(cls_dets, **cfg)
        else:
            cls_dets, _ = nms(cls_dets, **cfg)
        bboxes.append(cls_dets)
        labels.append(multi_bboxes.new_full((cls_dets.shape[0],), i - 1, dtype=torch.long))
    if not bboxes:
        return multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0,), dtype=torch.long)
    bboxes, labels = torch.cat(bboxes), torch.cat(labels)
    if bboxes.shape[0] > max_num:
        # (torch.sort on CPU / device is not stable by contract; the canonical order of this build:
        # equal scores keep their concatenation order)
        inds = torch.sort(bboxes[:, -1], descending=True, stable=True)[1][:max_num]
        bboxes, labels = bboxes[inds], labels[inds]
    return bboxes, labels
