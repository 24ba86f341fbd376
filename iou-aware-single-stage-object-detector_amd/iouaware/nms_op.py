"""`mmdet.ops.nms.nms` and `mmdet.core.multiclass_nms` with the reference's
call signatures, on the HIP kernels (reference mmdet/ops/nms/nms_wrapper.py:8-49,
mmdet/core/post_processing/bbox_nms.py:6-67)."""
import numpy as np
import torch

from . import ops


def nms(dets, iou_thr, device_id=None):
    """dets: (n,5) Tensor on a ROCm device, or ndarray + device_id.  Returns (dets[inds], inds)
    in the input's type; inds ascending (the CPU reference's order, nms_cpu.cpp:58).

    The reference dispatches CPU tensors to its C++ loop; this build has no CPU compute
    path -- a CPU tensor or an ndarray without device_id raises.
    """
    if isinstance(dets, torch.Tensor):
        is_numpy, dets_th = False, dets
    elif isinstance(dets, np.ndarray):
        is_numpy = True
        if device_id is None:
            raise ops._lib.IouAwareLibraryError(
                'nms on a numpy array needs device_id: there is no CPU NMS in this build')
        dets_th = torch.from_numpy(dets).to('cuda:{}'.format(device_id))
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(
            type(dets)))
    if dets_th.shape[0] == 0:
        inds = dets_th.new_zeros(0, dtype=torch.long)
    else:
        inds = ops.nms_indices(dets_th, iou_thr)
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
    cap = ops._lib.IA_MAX_PER_IMG
    if n > ops._lib.IA_MAX_CANDIDATES:
        raise ValueError('multiclass_nms: %d boxes, the HIP path handles at most %d per image'
                         % (n, ops._lib.IA_MAX_CANDIDATES))
    drop_last = max_num is None or max_num < 0
    if drop_last:
        # reference quirk (bbox_nms.py:52-56): `shape[0] > -1` is always true, so ALL survivors
        # are sorted by score (descending) and `inds[:-1]` drops the globally lowest one.
        # Emulated while the survivor count fits the library's per-image output buffer: the
        # kernel is asked for `cap` rows, which it returns unsorted (class-major) when fewer
        # survive; the sort and the drop happen below.
        max_num = cap
    elif max_num > cap:
        raise ValueError('multiclass_nms: max_num=%d exceeds the per-image output capacity %d'
                         % (max_num, cap))
    Cn = multi_scores.shape[1] - 1
    Rs = (n + 63) // 64 * 64
    scores_t = multi_bboxes.new_zeros((1, Cn, Rs), dtype=torch.float32)
    scores_t[0, :, :n] = multi_scores[:, 1:].t().to(torch.float32)
    boxes = multi_bboxes.to(torch.float32).reshape(1, n, 4)
    if nms_type == 'soft_nms':
        iou_thr = cfg.pop('iou_thr')
        out = ops.multiclass_soft_nms(boxes, scores_t, n, score_thr, iou_thr, int(max_num), **cfg)
    else:
        out = ops.multiclass_nms(boxes, scores_t, n, score_thr, cfg['iou_thr'], int(max_num))
    k = int(out[3][0].item())
    if drop_last:
        if k >= cap:
            raise ValueError('multiclass_nms(max_num=-1): %d or more survivors, beyond the '
                             'per-image output capacity; pass an explicit max_num' % cap)
        dets, labels = out[0][0, :k], out[1][0, :k].to(torch.long)
        # stable: equal scores keep their concatenation order (class ascending, row ascending)
        order = torch.sort(dets[:, 4], descending=True, stable=True)[1][:max(k - 1, 0)]
        return dets[order], labels[order]
    return out[0][0, :k], out[1][0, :k].to(torch.long)
