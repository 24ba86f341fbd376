"""Training-target assignment: MaxIoU assignment,
pseudo sampling, bbox2delta, per-level regrouping.  Torch ops on whatever device
the anchors live on; the next hot-spot to move into HIP after the loss kernels
(SURVEY 8f.2).  Follows reference mmdet/core/anchor/anchor_target.py:7-282,
mmdet/core/bbox/assigners/max_iou_assigner.py:50-201,
mmdet/core/bbox/samplers/pseudo_sampler.py:18-38.
"""
import torch

from .bbox import bbox2delta, bbox_overlaps


class AssignResult(object):
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = \
            num_gts, gt_inds, max_overlaps, labels


class BaseAssigner(object):
    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        raise NotImplementedError


class MaxIoUAssigner(BaseAssigner):
    """-1 ignore / 0 negative / k>0 assigned to gt k-1.  neg: max IoU < neg_iou_thr;
    pos: max IoU >= pos_iou_thr; every gt additionally claims the anchor(s) with its
    highest IoU when that IoU >= min_pos_iou (all of them if gt_max_assign_all)."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True,
                 ignore_iof_thr=-1, ignore_wrt_candidates=True):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr, self.ignore_wrt_candidates = ignore_iof_thr, ignore_wrt_candidates

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        if bboxes.shape[0] == 0 or gt_bboxes.shape[0] == 0:
            raise ValueError('No gt or bboxes')
        bboxes = bboxes[:, :4]
        overlaps = bbox_overlaps(gt_bboxes, bboxes)
        if self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0:
            if self.ignore_wrt_candidates:
                ig = bbox_overlaps(bboxes, gt_bboxes_ignore, mode='iof').max(dim=1)[0]
            else:
                ig = bbox_overlaps(gt_bboxes_ignore, bboxes, mode='iof').max(dim=0)[0]
            overlaps[:, ig > self.ignore_iof_thr] = -1
        return self.assign_wrt_overlaps(overlaps, gt_labels)

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        if overlaps.numel() == 0:
            raise ValueError('No gt or proposals')
        num_gts, num_bboxes = overlaps.size(0), overlaps.size(1)
        assigned = overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
        max_ov, argmax_ov = overlaps.max(dim=0)
        gt_max_ov, gt_argmax_ov = overlaps.max(dim=1)
        if isinstance(self.neg_iou_thr, float):
            assigned[(max_ov >= 0) & (max_ov < self.neg_iou_thr)] = 0
        elif isinstance(self.neg_iou_thr, tuple):
            lo, hi = self.neg_iou_thr
            assigned[(max_ov >= lo) & (max_ov < hi)] = 0
        pos = max_ov >= self.pos_iou_thr
        assigned[pos] = argmax_ov[pos] + 1
        for i in range(num_gts):           # later gts overwrite earlier ones, as the reference
            if gt_max_ov[i] >= self.min_pos_iou:
                if self.gt_max_assign_all:
                    assigned[overlaps[i, :] == gt_max_ov[i]] = i + 1
                else:
                    assigned[gt_argmax_ov[i]] = i + 1
        labels = None
        if gt_labels is not None:
            labels = assigned.new_zeros((num_bboxes,))
            pos_inds = torch.nonzero(assigned > 0).squeeze(-1)
            if pos_inds.numel() > 0:
                labels[pos_inds] = gt_labels[assigned[pos_inds] - 1]
        return AssignResult(num_gts, assigned, max_ov, labels=labels)


class SamplingResult(object):
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = None if assign_result.labels is None else assign_result.labels[pos_inds]


class PseudoSampler(object):
    """no sampling (focal loss trains on every anchor): all positives, all negatives."""

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos = torch.nonzero(assign_result.gt_inds > 0).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0).squeeze(-1).unique()
        flags = bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8)
        return SamplingResult(pos, neg, bboxes, gt_bboxes, assign_result, flags)


_ASSIGNERS = {'MaxIoUAssigner': MaxIoUAssigner}


def build_assigner(cfg, **kwargs):
    if isinstance(cfg, BaseAssigner):
        return cfg
    if not isinstance(cfg, dict):
        raise TypeError('Invalid type {} for building an assigner'.format(type(cfg)))
    args = dict(cfg)
    kind = args.pop('type')
    if isinstance(kind, str):
        if kind not in _ASSIGNERS:
            raise KeyError('{} is not a known assigner'.format(kind))
        kind = _ASSIGNERS[kind]
    for k, v in kwargs.items():
        args.setdefault(k, v)
    return kind(**args)


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    if allowed_border < 0:
        return valid_flags
    h, w = img_shape[:2]
    return valid_flags & (flat_anchors[:, 0] >= -allowed_border).type(torch.uint8) & \
        (flat_anchors[:, 1] >= -allowed_border).type(torch.uint8) & \
        (flat_anchors[:, 2] < w + allowed_border).type(torch.uint8) & \
        (flat_anchors[:, 3] < h + allowed_border).type(torch.uint8)


def unmap(data, count, inds, fill=0):
    if data.dim() == 1:
        ret = data.new_full((count,), fill)
        ret[inds] = data
    else:
        ret = data.new_full((count,) + data.size()[1:], fill)
        ret[inds, :] = data
    return ret


def expand_binary_labels(labels, label_weights, label_channels):
    """(N,) labels in 0..C -> (N,C) one-hot int64 and the (N,C) expanded weight view."""
    bin_labels = labels.new_full((labels.size(0), label_channels), 0)
    inds = torch.nonzero(labels >= 1).squeeze(-1)
    if inds.numel() > 0:
        bin_labels[inds, labels[inds] - 1] = 1
    return bin_labels, label_weights.view(-1, 1).expand(label_weights.size(0), label_channels)


def images_to_levels(target, num_level_anchors):
    target = torch.stack(target, 0)
    out, start = [], 0
    for n in num_level_anchors:
        out.append(target[:, start:start + n].squeeze(0))
        start += n
    return out


def anchor_target_single(flat_anchors, valid_flags, gt_bboxes, gt_bboxes_ignore, gt_labels, img_meta,
                         target_means, target_stds, cfg, label_channels=1, sampling=True,
                         unmap_outputs=True):
    inside = anchor_inside_flags(flat_anchors, valid_flags, img_meta['img_shape'][:2],
                                 cfg.allowed_border)
    if not inside.any():
        return (None,) * 6
    keep = inside.bool()
    anchors = flat_anchors[keep, :]
    if sampling:
        raise NotImplementedError('sampled (RPN-style) targets are outside this build: the '
                                  'IoU-aware RetinaNet configs use focal loss (sampling=False)')
    assigner = build_assigner(cfg.assigner)
    assign_result = assigner.assign(anchors, gt_bboxes, gt_bboxes_ignore, gt_labels)
    sampled = PseudoSampler().sample(assign_result, anchors, gt_bboxes)
    pos_inds, neg_inds = sampled.pos_inds, sampled.neg_inds
    n_valid = anchors.shape[0]
    bbox_targets = torch.zeros_like(anchors)
    bbox_weights = torch.zeros_like(anchors)
    labels = anchors.new_zeros(n_valid, dtype=torch.long)
    label_weights = anchors.new_zeros(n_valid, dtype=torch.float)
    if len(pos_inds) > 0:
        bbox_targets[pos_inds, :] = bbox2delta(sampled.pos_bboxes, sampled.pos_gt_bboxes,
                                               target_means, target_stds)
        bbox_weights[pos_inds, :] = 1.0
        labels[pos_inds] = 1 if gt_labels is None else gt_labels[sampled.pos_assigned_gt_inds]
        label_weights[pos_inds] = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
    if len(neg_inds) > 0:
        label_weights[neg_inds] = 1.0
    if unmap_outputs:
        total = flat_anchors.size(0)
        labels = unmap(labels, total, keep)
        label_weights = unmap(label_weights, total, keep)
        bbox_targets = unmap(bbox_targets, total, keep)
        bbox_weights = unmap(bbox_weights, total, keep)
    return labels, label_weights, bbox_targets, bbox_weights, pos_inds, neg_inds


def anchor_target(anchor_list, valid_flag_list, gt_bboxes_list, img_metas, target_means,
                  target_stds, cfg, gt_bboxes_ignore_list=None, gt_labels_list=None,
                  label_channels=1, sampling=True, unmap_outputs=True):
    """-> (labels[L], label_weights[L], bbox_targets[L], bbox_weights[L], num_total_pos,
    num_total_neg, level_anchors[L]) or None when an image has no valid anchor.
    num_total_pos = sum over images of max(n_pos, 1)  (anchor_target.py:94)."""
    num_imgs = len(img_metas)
    if not len(anchor_list) == len(valid_flag_list) == num_imgs:
        raise AssertionError('per-image anchor lists do not match the batch')
    num_level_anchors = [a.size(0) for a in anchor_list[0]]
    flat_anchors = [torch.cat(a) for a in anchor_list]
    flat_flags = [torch.cat(f) for f in valid_flag_list]
    ignore = gt_bboxes_ignore_list or [None] * num_imgs
    gt_labels = gt_labels_list or [None] * num_imgs
    per_img = [anchor_target_single(flat_anchors[i], flat_flags[i], gt_bboxes_list[i], ignore[i],
                                    gt_labels[i], img_metas[i], target_means, target_stds, cfg,
                                    label_channels=label_channels, sampling=sampling,
                                    unmap_outputs=unmap_outputs) for i in range(num_imgs)]
    if any(r[0] is None for r in per_img):
        return None
    labels, weights, btargets, bweights, pos, neg = map(list, zip(*per_img))
    num_total_pos = sum(max(p.numel(), 1) for p in pos)
    num_total_neg = sum(max(n.numel(), 1) for n in neg)
    return (images_to_levels(labels, num_level_anchors),
            images_to_levels(weights, num_level_anchors),
            images_to_levels(btargets, num_level_anchors),
            images_to_levels(bweights, num_level_anchors), num_total_pos, num_total_neg,
            images_to_levels(flat_anchors, num_level_anchors))
