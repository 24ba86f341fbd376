"""Box codec / IoU helpers in torch, with the reference's "+1 pixel" convention
(reference mmdet/core/bbox/transforms.py:6-78,148-166; geometry.py:4-66).

On the inference and loss hot paths these are fused into HIP kernels
(csrc/decode.hip, csrc/loss.hip); the torch versions serve the training-target
assignment ("stays PyTorch") and API parity.
"""
import numpy as np
import torch

_MAX_RATIO_DEFAULT = 16 / 1000


def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    if proposals.size() != gt.size():
        raise AssertionError('proposals / gt shape mismatch')
    p, g = proposals.float(), gt.float()
    pw = p[..., 2] - p[..., 0] + 1.0
    ph = p[..., 3] - p[..., 1] + 1.0
    gw = g[..., 2] - g[..., 0] + 1.0
    gh = g[..., 3] - g[..., 1] + 1.0
    dx = ((g[..., 0] + g[..., 2]) * 0.5 - (p[..., 0] + p[..., 2]) * 0.5) / pw
    dy = ((g[..., 1] + g[..., 3]) * 0.5 - (p[..., 1] + p[..., 3]) * 0.5) / ph
    deltas = torch.stack([dx, dy, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    m = deltas.new_tensor(means).unsqueeze(0)
    s = deltas.new_tensor(stds).unsqueeze(0)
    return deltas.sub_(m).div_(s)


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=_MAX_RATIO_DEFAULT):
    reps = deltas.size(1) // 4
    m = deltas.new_tensor(means).repeat(1, reps)
    s = deltas.new_tensor(stds).repeat(1, reps)
    d = deltas * s + m
    dx, dy, dw, dh = d[:, 0::4], d[:, 1::4], d[:, 2::4], d[:, 3::4]
    lim = float(np.abs(np.log(wh_ratio_clip)))
    dw = dw.clamp(min=-lim, max=lim)
    dh = dh.clamp(min=-lim, max=lim)
    px = ((rois[:, 0] + rois[:, 2]) * 0.5).unsqueeze(1).expand_as(dx)
    py = ((rois[:, 1] + rois[:, 3]) * 0.5).unsqueeze(1).expand_as(dy)
    pw = (rois[:, 2] - rois[:, 0] + 1.0).unsqueeze(1).expand_as(dw)
    ph = (rois[:, 3] - rois[:, 1] + 1.0).unsqueeze(1).expand_as(dh)
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx, gy = px + pw * dx, py + ph * dy
    x1, y1 = gx - gw * 0.5 + 0.5, gy - gh * 0.5 + 0.5
    x2, y2 = gx + gw * 0.5 - 0.5, gy + gh * 0.5 - 0.5
    if max_shape is not None:
        x1 = x1.clamp(min=0, max=max_shape[1] - 1)
        y1 = y1.clamp(min=0, max=max_shape[0] - 1)
        x2 = x2.clamp(min=0, max=max_shape[1] - 1)
        y2 = y2.clamp(min=0, max=max_shape[0] - 1)
    return torch.stack([x1, y1, x2, y2], dim=-1).view_as(deltas)


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False):
    """(m,n) IoU / IoF matrix, or (m,) aligned pairs; widths are x2-x1+1."""
    if mode not in ('iou', 'iof'):
        raise AssertionError(mode)
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if is_aligned and rows != cols:
        raise AssertionError('aligned overlaps need equal counts')
    if rows * cols == 0:
        return bboxes1.new(rows, 1) if is_aligned else bboxes1.new(rows, cols)
    area1 = (bboxes1[:, 2] - bboxes1[:, 0] + 1) * (bboxes1[:, 3] - bboxes1[:, 1] + 1)
    area2 = (bboxes2[:, 2] - bboxes2[:, 0] + 1) * (bboxes2[:, 3] - bboxes2[:, 1] + 1)
    if is_aligned:
        lt = torch.max(bboxes1[:, :2], bboxes2[:, :2])
        rb = torch.min(bboxes1[:, 2:], bboxes2[:, 2:])
        wh = (rb - lt + 1).clamp(min=0)
        inter = wh[:, 0] * wh[:, 1]
        return inter / (area1 + area2 - inter) if mode == 'iou' else inter / area1
    lt = torch.max(bboxes1[:, None, :2], bboxes2[:, :2])
    rb = torch.min(bboxes1[:, None, 2:], bboxes2[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    if mode == 'iou':
        return inter / (area1[:, None] + area2 - inter)
    return inter / area1[:, None]


def bbox2result(bboxes, labels, num_classes):
    """(n,5) + (n,) -> list of num_classes-1 ndarrays (k_c,5) fp32 (transforms.py:148-166)."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes - 1)]
    b = bboxes.detach().cpu().numpy() if torch.is_tensor(bboxes) else np.asarray(bboxes)
    l = labels.detach().cpu().numpy() if torch.is_tensor(labels) else np.asarray(labels)
    return [b[l == i, :] for i in range(num_classes - 1)]


def multi_apply(func, *args, **kwargs):
    """map func over per-level argument lists, transpose the results into a tuple of lists
    (reference mmdet/core/utils/misc.py:21-24)."""
    from functools import partial
    f = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(f, *args))))
