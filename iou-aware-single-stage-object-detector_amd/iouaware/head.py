"""IoU-aware RetinaNet head (reference
mmdet/models/anchor_heads/iou_aware_retina_head.py:64-564 on top of
anchor_head.py:21-148), with the reference's registry name, constructor
kwargs, parameter names and method signatures.

What is different from the reference is WHERE the work after the last
convolution happens:
  * forward / forward_single: the 4+4 conv towers and the three output convs
    stay PyTorch-ROCm modules (dense contractions -> MIOpen on MFMA);
  * get_bboxes: one call into the HIP library for the whole batch (row-max,
    per-level top-k, gather/decode, batched NMS, final top-k) instead of ~25
    eager ops per level and 80 NMS calls per image;
  * loss: targets from torch (targets.py), losses from the HIP kernels on the
    NCHW tensors directly.
There is no CPU fallback: get_bboxes / loss need tensors on a gfx950 device.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .anchors import AnchorGenerator
from .bbox import multi_apply
from .layers import ConvModule, bias_init_with_prob, normal_init
from .registry import HEADS, build_loss
from .targets import anchor_target

_NO_SAMPLING_LOSSES = ('FocalLoss', 'GHMC', 'IOUbalancedSigmoidFocalLoss')


@HEADS.register_module
class AnchorHead(nn.Module):
    """Anchor bookkeeping shared by anchor-based heads: generators per level,
    `forward = multi_apply(forward_single)`, `get_anchors`
    (reference anchor_head.py:37-148).  The plain 1x1-conv RPN-style losses /
    decoding of the reference base class are outside the IoU-aware path."""

    def __init__(self, num_classes, in_channels, feat_channels=256, anchor_scales=[8, 16, 32],
                 anchor_ratios=[0.5, 1.0, 2.0], anchor_strides=[4, 8, 16, 32, 64],
                 anchor_base_sizes=None, target_means=(.0, .0, .0, .0),
                 target_stds=(1.0, 1.0, 1.0, 1.0),
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0)):
        super(AnchorHead, self).__init__()
        self.in_channels, self.num_classes, self.feat_channels = in_channels, num_classes, feat_channels
        self.anchor_scales, self.anchor_ratios = anchor_scales, anchor_ratios
        self.anchor_strides = anchor_strides
        self.anchor_base_sizes = list(anchor_strides) if anchor_base_sizes is None \
            else anchor_base_sizes
        self.target_means, self.target_stds = target_means, target_stds
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        self.sampling = loss_cls['type'] not in _NO_SAMPLING_LOSSES
        self.cls_out_channels = num_classes - 1 if self.use_sigmoid_cls else num_classes
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.anchor_generators = [AnchorGenerator(b, anchor_scales, anchor_ratios)
                                  for b in self.anchor_base_sizes]
        self.num_anchors = len(self.anchor_ratios) * len(self.anchor_scales)
        self.IoU_balanced_Cls = loss_cls['type'] in ('IOUbalancedCrossEntropyLoss',
                                                     'IOUbalancedSigmoidFocalLoss')
        self.IoU_balanced_Loc = loss_bbox['type'] in ('IoUbalancedSmoothL1Loss',)
        self._geom_cache = {}
        self._init_layers()

    def _init_layers(self):
        self.conv_cls = nn.Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 1)
        self.conv_reg = nn.Conv2d(self.feat_channels, self.num_anchors * 4, 1)

    def init_weights(self):
        normal_init(self.conv_cls, std=0.01)
        normal_init(self.conv_reg, std=0.01)

    def forward_single(self, x):
        return self.conv_cls(x), self.conv_reg(x)

    def forward(self, feats):
        return multi_apply(self.forward_single, feats)

    def get_anchors(self, featmap_sizes, img_metas, device='cpu'):
        """anchors once per batch, valid flags per image from pad_shape
        (reference anchor_head.py:105-148)."""
        levels = [self.anchor_generators[i].grid_anchors(featmap_sizes[i], self.anchor_strides[i],
                                                         device=device)
                  for i in range(len(featmap_sizes))]
        anchor_list = [list(levels) for _ in img_metas]
        valid_flag_list = []
        for meta in img_metas:
            h, w = meta['pad_shape'][:2]
            flags = []
            for i, (fh, fw) in enumerate(featmap_sizes):
                s = self.anchor_strides[i]
                vh = min(int(np.ceil(h / s)), int(fh))
                vw = min(int(np.ceil(w / s)), int(fw))
                flags.append(self.anchor_generators[i].valid_flags((fh, fw), (vh, vw),
                                                                   device=device))
            valid_flag_list.append(flags)
        return anchor_list, valid_flag_list

    def geometry(self, featmap_sizes, nms_pre=-1):
        """ia_head_geom for these feature-map sizes (cached)."""
        key = (tuple(tuple(int(v) for v in s) for s in featmap_sizes), int(nms_pre))
        g = self._geom_cache.get(key)
        if g is None:
            base = np.stack([gen.base_anchors.numpy() for gen in self.anchor_generators])
            # score columns = foreground classes; a softmax head (use_sigmoid_cls=False) carries
            # num_classes channels per anchor with the background in channel 0 (:506-507,540-541)
            g = ops.HeadGeometry(key[0], self.anchor_strides, base, self.num_classes - 1,
                                 nms_pre=nms_pre, means=self.target_means, stds=self.target_stds,
                                 softmax=not self.use_sigmoid_cls)
            self._geom_cache[key] = g
        return g


@HEADS.register_module
class IoUawareRetinaHead(AnchorHead):
    """RetinaNet head with a class-agnostic IoU branch: `retina_iou` (A channels) reads
    the regression tower's last feature; detection confidence is
    sigmoid(cls)^0.5 * sigmoid(iou)^0.5 (alpha = 0.5 hard-coded in the reference, :510)."""

    score_alpha = 0.5

    def __init__(self, num_classes, in_channels, stacked_convs=4, octave_base_scale=4,
                 scales_per_octave=3, conv_cfg=None, norm_cfg=None,
                 loss_iou=dict(type='GHMIoU', bins=30, momentum=0.75, use_sigmoid=True,
                               loss_weight=1.0),
                 attach_iou_target=True, **kwargs):
        self.stacked_convs = stacked_convs
        self.octave_base_scale, self.scales_per_octave = octave_base_scale, scales_per_octave
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        # `loss_iou` is accepted and ignored, exactly like the reference (:74, never used).
        self.attach_iou_target = attach_iou_target
        octave_scales = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
        super(IoUawareRetinaHead, self).__init__(
            num_classes, in_channels, anchor_scales=octave_scales * octave_base_scale, **kwargs)

    def _init_layers(self):
        self.relu = nn.ReLU(inplace=True)
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            ch = self.in_channels if i == 0 else self.feat_channels
            for tower in (self.cls_convs, self.reg_convs):
                tower.append(ConvModule(ch, self.feat_channels, 3, stride=1, padding=1,
                                        conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
        self.retina_cls = nn.Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 3,
                                    padding=1)
        self.retina_reg = nn.Conv2d(self.feat_channels, self.num_anchors * 4, 3, padding=1)
        self.shared_conv = 4              # IoU branch shares all four regression convs
        self.use_feature_alignment = False
        self.retina_iou = nn.Conv2d(self.feat_channels, self.num_anchors, 3, padding=1)

    def init_weights(self):
        for m in list(self.cls_convs) + list(self.reg_convs):
            normal_init(m.conv, std=0.01)
        normal_init(self.retina_cls, std=0.01, bias=bias_init_with_prob(0.01))
        normal_init(self.retina_reg, std=0.01)
        normal_init(self.retina_iou, std=0.01)

    train_winograd = True                 # training: all-levels Winograd convolutions when usable

    def forward(self, feats):
        """multi_apply(forward_single) of the reference (anchor_head.py:102-103); in training on a
        ROCm device every convolution runs once for all levels on the Winograd path with its own
        backward (iouaware/winograd_train.py) -- same parameters, same outputs to fp32 rounding."""
        if self.training and self.train_winograd:
            from . import winograd_train
            if winograd_train.usable(feats, self):
                return winograd_train.head_forward(self, feats)
        return super(IoUawareRetinaHead, self).forward(feats)

    def forward_single(self, x):
        cls_feat = reg_feat = x
        for conv in self.cls_convs:
            cls_feat = conv(cls_feat)
        for conv in self.reg_convs:
            reg_feat = conv(reg_feat)
        return self.retina_cls(cls_feat), self.retina_reg(reg_feat), self.retina_iou(reg_feat)

    # ------------------------------------------------------------------ inference
    def get_bboxes_batched(self, cls_scores, bbox_preds, iou_preds, img_metas, cfg, rescale=False):
        """Device-side result of the whole batch: dets (B,max,5), labels (B,max) int32,
        rows (B,max) int32, num (B) int32 -- no host synchronisation."""
        if not len(cls_scores) == len(bbox_preds) == len(iou_preds) == len(self.anchor_generators):
            raise AssertionError('level count mismatch')
        nms_cfg = dict(cfg.nms)
        nms_type = nms_cfg.pop('type', 'nms')
        if nms_type not in ('nms', 'soft_nms'):
            raise AttributeError("module 'nms_wrapper' has no attribute '%s'" % nms_type)
        iou_thr = nms_cfg.pop('iou_thr')
        soft = nms_cfg if nms_type == 'soft_nms' else None      # method / sigma / min_score
        featmap_sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        geom = self.geometry(featmap_sizes, cfg.get('nms_pre', -1))
        shapes = [m['img_shape'] for m in img_metas]
        factors = [m['scale_factor'] for m in img_metas]
        cls_scores = [c.detach() for c in cls_scores]
        bbox_preds = [b.detach() for b in bbox_preds]
        iou_preds = [i.detach() for i in iou_preds]
        return ops.get_bboxes(geom, cls_scores, bbox_preds, iou_preds, shapes, factors, rescale,
                              cfg.score_thr, iou_thr, cfg.max_per_img, soft=soft)

    def get_bboxes(self, cls_scores, bbox_preds, iou_preds, gt_bboxes, gt_labels, img_metas, cfg,
                   rescale=False):
        """-> list over images of (det_bboxes (k,5) fp32, det_labels (k,) int64), k <= max_per_img.
        gt_bboxes / gt_labels are accepted positionally like the fork's signature (:390-398);
        they only feed dead code there (:517-524) and are unused."""
        dets, labels, _, num = self.get_bboxes_batched(cls_scores, bbox_preds, iou_preds, img_metas,
                                                       cfg, rescale)
        counts = num.tolist()                                   # the one host sync per batch
        return [(dets[b, :k], labels[b, :k].to(torch.long)) for b, k in enumerate(counts)]

    # ------------------------------------------------------------------ training
    def loss_single(self, cls_score, bbox_pred, iou_pred, labels, label_weights, bbox_targets,
                    bbox_weights, level_anchor, num_total_samples, gt_bboxes, cfg, level=None,
                    geom=None):
        """losses of one pyramid level; each a (1,)-shaped tensor (reference :221-313).
        `level_anchor` is accepted for signature parity: anchors are regenerated in-kernel."""
        B = cls_score.shape[0]
        if geom is None:
            raise AssertionError('loss_single needs the level geometry')
        n_l = geom.level_anchors[level]
        bbox_targets, bbox_weights = bbox_targets.reshape(B, n_l, 4), bbox_weights.reshape(B, n_l, 4)
        labels, label_weights = labels.reshape(B, n_l), label_weights.reshape(B, n_l)
        balanced = self.IoU_balanced_Cls or self.IoU_balanced_Loc
        # the IoU regression target (:256-259) also weights the IoU-balanced losses (detached
        # there: losses.py:356,448)
        out = ops.iou_bce_sum(bbox_pred, iou_pred, bbox_targets, bbox_weights, geom, level,
                              self.attach_iou_target, return_iou=balanced)
        loss_iou, iou = (out if balanced else (out, None))
        loss_iou = loss_iou * (1.0 / num_total_samples)
        if self.IoU_balanced_Loc:
            loss_bbox = self.loss_bbox.forward_level(bbox_pred, bbox_targets, bbox_weights, iou,
                                                     self.num_anchors, num_total_samples)
        else:
            loss_bbox = self.loss_bbox.forward_level(bbox_pred, bbox_targets, bbox_weights,
                                                     self.num_anchors, num_total_samples)
        if self.IoU_balanced_Cls:
            loss_cls = self.loss_cls.forward_level(cls_score, labels, label_weights, iou,
                                                   self.num_anchors, num_total_samples)
        else:
            loss_cls = self.loss_cls.forward_level(cls_score, labels, label_weights,
                                                   self.num_anchors, num_total_samples)
        return loss_cls, loss_bbox, loss_iou

    def _device_targets_ok(self, cfg, gt_bboxes, gt_bboxes_ignore, device):
        """the HIP assigner covers the IoU-aware configs' train_cfg (MaxIoUAssigner,
        gt_max_assign_all, allowed_border=-1, no ignore regions, no sampling); anything else
        takes the torch path in targets.py"""
        a = cfg.assigner
        return (device.type == 'cuda' and not self.sampling and isinstance(a, dict)
                and a.get('type') == 'MaxIoUAssigner' and a.get('gt_max_assign_all', True)
                and isinstance(a.get('neg_iou_thr'), float) and cfg.allowed_border < 0
                and not (a.get('ignore_iof_thr', -1) > 0 and gt_bboxes_ignore is not None)
                and all(g.shape[0] >= 1 for g in gt_bboxes)
                and max(g.shape[0] for g in gt_bboxes) <= 512)

    fuse_levels = True                    # all-levels loss kernels when the configuration allows

    def _fused_loss_ok(self, cls_scores):
        from .losses import FocalLoss, SmoothL1Loss
        return (self.fuse_levels and type(self.loss_cls) is FocalLoss
                and type(self.loss_bbox) is SmoothL1Loss and float(self.loss_cls.gamma) == 2.0
                and self.use_sigmoid_cls and not self.sampling and cls_scores[0].is_cuda
                and cls_scores[0].dtype in (torch.float32, torch.bfloat16))

    def loss(self, cls_scores, bbox_preds, iou_preds, gt_bboxes, gt_labels, img_metas, cfg,
             gt_bboxes_ignore=None):
        """-> dict(loss_cls, loss_bbox, losses_iou), each a list of per-level (1,) tensors, or
        None when an image has no valid anchor (reference :315-387; the key really is
        'losses_iou', :387)."""
        featmap_sizes = [tuple(f.shape[-2:]) for f in cls_scores]
        if len(featmap_sizes) != len(self.anchor_generators):
            raise AssertionError('level count mismatch')
        device = cls_scores[0].device
        geom = self.geometry(featmap_sizes, -1)
        if self._device_targets_ok(cfg, gt_bboxes, gt_bboxes_ignore, device):
            # whole batch in two HIP launches, nothing returns to the host: the normaliser
            # num_total_pos = sum_i max(n_pos_i, 1) (anchor_target.py:94) stays a device scalar
            acfg = cfg.assigner
            labels, label_w, bbox_t, bbox_w, counts = ops.anchor_targets(
                geom, gt_bboxes, gt_labels, [m['pad_shape'] for m in img_metas],
                acfg['pos_iou_thr'], acfg['neg_iou_thr'], acfg.get('min_pos_iou', .0),
                cfg.pos_weight)
            level_anchors = [None] * len(featmap_sizes)
            if self._fused_loss_ok(cls_scores):
                num_total_samples = counts             # reduced inside the finalize kernel
            else:
                num_total_samples = counts[:, 0].clamp(min=1).sum().to(torch.float32)
        else:
            anchor_list, valid_flag_list = self.get_anchors(featmap_sizes, img_metas, device=device)
            label_channels = self.cls_out_channels if self.use_sigmoid_cls else 1
            targets = anchor_target(anchor_list, valid_flag_list, gt_bboxes, img_metas,
                                    self.target_means, self.target_stds, cfg,
                                    gt_bboxes_ignore_list=gt_bboxes_ignore,
                                    gt_labels_list=gt_labels, label_channels=label_channels,
                                    sampling=self.sampling)
            if targets is None:
                return None
            labels, label_w, bbox_t, bbox_w, n_pos, n_neg, level_anchors = targets
            num_total_samples = n_pos + n_neg if self.sampling else n_pos
            counts = None
        if self._fused_loss_ok(cls_scores):
            # all levels, all three losses: one autograd node, 3 + 2 kernel launches
            # (csrc/headloss.hip); the per-level path below stays for the other loss types
            on_dev = level_anchors[0] is None          # targets came from the HIP assigner
            return ops.head_loss(
                geom, cls_scores, bbox_preds, iou_preds, labels, label_w, bbox_t, bbox_w,
                counts=counts if on_dev else None,
                avg_factor=None if on_dev else num_total_samples,
                gamma=self.loss_cls.gamma, alpha=self.loss_cls.alpha,
                loss_weight_cls=self.loss_cls.loss_weight, beta=self.loss_bbox.beta,
                loss_weight_bbox=self.loss_bbox.loss_weight,
                attach_iou_target=self.attach_iou_target)
        out = [self.loss_single(cls_scores[l], bbox_preds[l], iou_preds[l], labels[l], label_w[l],
                                bbox_t[l], bbox_w[l], level_anchors[l], num_total_samples,
                                gt_bboxes, cfg, level=l, geom=geom)
               for l in range(len(featmap_sizes))]
        losses_cls, losses_bbox, losses_iou = map(list, zip(*out))
        return dict(loss_cls=losses_cls, loss_bbox=losses_bbox, losses_iou=losses_iou)
