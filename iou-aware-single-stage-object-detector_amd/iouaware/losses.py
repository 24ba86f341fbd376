"""Loss modules the IoU-aware configs name (`FocalLoss`, `SmoothL1Loss`),
backed by the HIP loss kernels (csrc/loss.hip).

Two call forms:
  * the reference's module signature on permuted (N, C) tensors
    (reference mmdet/models/losses/focal_loss.py:23-35, smooth_l1_loss.py:14-18);
  * `forward_level(...)` on the NCHW head output of one pyramid level, which is
    what IoUawareRetinaHead.loss_single uses: no permute copy, no (N,80) int64
    one-hot, no (N,80) expanded weight.
Both return a (1,)-shaped tensor = loss_weight * sum / avg_factor, like the
reference's weighted_* functions (losses.py:279-303, 403-411).
"""
import torch
import torch.nn as nn

from . import ops
from .registry import LOSSES


@LOSSES.register_module
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=False, loss_weight=1.0, gamma=2.0, alpha=0.25):
        super(FocalLoss, self).__init__()
        if use_sigmoid is not True:
            raise AssertionError('Only sigmoid focaloss supported now.')
        self.use_sigmoid, self.loss_weight, self.gamma, self.alpha = \
            use_sigmoid, loss_weight, gamma, alpha

    def forward_level(self, cls_score, labels, label_weights, num_anchors, avg_factor):
        """cls_score (B, A*C, H, W); labels (B, N_l) int64 in 0..C; label_weights (B, N_l)."""
        total = ops.focal_loss_sum(cls_score, labels, label_weights, num_anchors, self.gamma,
                                   self.alpha)
        return total * (self.loss_weight / avg_factor)      # avg_factor: python number or device scalar

    def forward(self, cls_score, label, label_weight, avg_factor=None, **kwargs):
        """cls_score (N, C); label (N, C) one-hot or (N,) integer 0..C; label_weight (N, C)
        row-constant or (N,)."""
        if cls_score.dim() != 2:
            raise AssertionError('cls_score must be (N, C)')
        n, c = cls_score.shape
        if label.dim() == 2:
            hot = label > 0
            label = torch.where(hot.any(1), hot.to(torch.int64).argmax(1) + 1,
                                torch.zeros(n, dtype=torch.int64, device=label.device))
        if label_weight.dim() == 2:
            label_weight = label_weight[:, 0]
        if avg_factor is None:
            avg_factor = float((label_weight > 0).sum().item()) + 1e-6
        # an (N, C) row-major matrix is the NCHW layout with B=N, A=1, HW=1
        total = ops.focal_loss_sum(cls_score.reshape(n, c, 1, 1), label, label_weight, 1,
                                   self.gamma, self.alpha)
        return total * (self.loss_weight / avg_factor)      # avg_factor: python number or device scalar


@LOSSES.register_module
class CrossEntropyLoss(nn.Module):
    """The loss a softmax-classification head names (`loss_cls=dict(type='CrossEntropyLoss',
    use_sigmoid=False)`, reference mmdet/models/losses/cross_entropy_loss.py:9-31 with
    core/loss/losses.py:19-26,141-149).  Host-side torch arithmetic: none of the IoU-aware
    configs trains with it; it exists so that a use_sigmoid_cls=False head can be built (its
    INFERENCE branch, iou_aware_retina_head.py:506-507,540-541, runs on the HIP path).
    Returns a (1,)-shaped tensor = loss_weight * weighted sum / avg_factor."""

    def __init__(self, use_sigmoid=False, use_mask=False, loss_weight=1.0):
        super(CrossEntropyLoss, self).__init__()
        if use_mask:
            raise NotImplementedError('mask cross entropy belongs to the mask heads (out of scope)')
        self.use_sigmoid, self.use_mask, self.loss_weight = use_sigmoid, use_mask, loss_weight

    def forward_level(self, cls_score, labels, label_weights, num_anchors, avg_factor):
        """what AnchorHead.loss_single calls for a non-fused loss: cls_score (B, A*Cin, H, W), labels /
        label_weights (B, N_l) -> the reference's flattened rows, cls_score.permute(0, 2, 3, 1).reshape(-1, Cin)
        (anchor_head.py:150-160), then `forward` (the method was missing -- AttributeError on the
        first training step of a CrossEntropyLoss head)"""
        cin = cls_score.shape[1] // num_anchors
        rows = cls_score.permute(0, 2, 3, 1).reshape(-1, cin)
        return self.forward(rows, labels.reshape(-1), label_weights.reshape(-1), avg_factor=avg_factor)

    def forward(self, cls_score, label, label_weight, avg_factor=None, **kwargs):
        import torch.nn.functional as F
        if self.use_sigmoid:
            if cls_score.dim() != label.dim():            # integer labels 0..C -> one-hot over C columns
                hot = torch.zeros_like(cls_score)
                fg = torch.nonzero(label >= 1).flatten()
                hot[fg, label[fg] - 1] = 1.0
                label, label_weight = hot, label_weight.reshape(-1, 1).expand_as(cls_score)
            # the default normaliser counts the EXPANDED (N, C) weights, like weighted_binary_cross_entropy
            # (core/loss/losses.py:142-145: expansion first) -- ADVICE r5
            if avg_factor is None:
                avg_factor = max(float((label_weight > 0).sum().item()), 1.0)
            total = F.binary_cross_entropy_with_logits(cls_score, label.to(cls_score.dtype),
                                                       label_weight.to(cls_score.dtype), reduction='sum')
        else:
            if avg_factor is None:
                avg_factor = max(float((label_weight > 0).sum().item()), 1.0)
            total = (F.cross_entropy(cls_score, label, reduction='none') * label_weight).sum()
        return self.loss_weight * total.reshape(1) / avg_factor


@LOSSES.register_module
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, loss_weight=1.0):
        super(SmoothL1Loss, self).__init__()
        self.beta, self.loss_weight = beta, loss_weight

    def forward_level(self, bbox_pred, bbox_targets, bbox_weights, num_anchors, avg_factor):
        """bbox_pred (B, A*4, H, W); targets / weights (B, N_l, 4)."""
        total = ops.smooth_l1_sum(bbox_pred, bbox_targets, bbox_weights, num_anchors, self.beta)
        return total * (self.loss_weight / avg_factor)      # avg_factor: python number or device scalar

    def forward(self, pred, target, weight, avg_factor=None, **kwargs):
        """pred / target / weight (N, 4)."""
        if pred.size() != target.size() or target.numel() == 0:
            raise AssertionError('pred / target shape mismatch or empty')
        if avg_factor is None:
            avg_factor = float((weight > 0).sum().item()) / 4 + 1e-6
        n = pred.shape[0]
        total = ops.smooth_l1_sum(pred.reshape(n, 4, 1, 1), target.reshape(n, 1, 4),
                                  weight.reshape(n, 1, 4), 1, self.beta)
        return total * (self.loss_weight / avg_factor)      # avg_factor: python number or device scalar


@LOSSES.register_module
class IOUbalancedSigmoidFocalLoss(nn.Module):
    """reference mmdet/models/losses/iou_balanced_sigmoid_focal_loss.py:8-59; like the reference
    the module does NOT apply loss_weight (the multiplication is commented out there, :31-42)."""

    def __init__(self, use_sigmoid=False, loss_weight=1.0, gamma=2.0, alpha=0.25, eta=1.0):
        super(IOUbalancedSigmoidFocalLoss, self).__init__()
        if use_sigmoid is not True:
            raise AssertionError('Only sigmoid focaloss supported now.')
        self.use_sigmoid, self.loss_weight, self.gamma, self.alpha, self.eta = \
            use_sigmoid, loss_weight, gamma, alpha, eta

    def forward_level(self, cls_score, labels, label_weights, iou, num_anchors, avg_factor):
        """iou (B*N_l): IoU of each anchor's predicted box with its target box (detached)."""
        total = ops.focal_loss_balanced_sum(cls_score, labels, label_weights, iou, num_anchors,
                                            self.gamma, self.alpha, self.eta)
        return total * (1.0 / avg_factor)

    def forward(self, cls_score, label, label_weight, iou, avg_factor=None, **kwargs):
        """cls_score (N, C); label one-hot (N, C) or integer (N,); label_weight; iou (N,)."""
        n, c = cls_score.shape
        if label.dim() == 2:
            hot = label > 0
            label = torch.where(hot.any(1), hot.to(torch.int64).argmax(1) + 1,
                                torch.zeros(n, dtype=torch.int64, device=label.device))
        if label_weight.dim() == 2:
            label_weight = label_weight[:, 0]
        if avg_factor is None:
            raise TypeError('avg_factor is required (the reference divides by it unconditionally, '
                            'losses.py:374)')
        total = ops.focal_loss_balanced_sum(cls_score.reshape(n, c, 1, 1), label, label_weight, iou,
                                            1, self.gamma, self.alpha, self.eta)
        return total * (1.0 / avg_factor)


@LOSSES.register_module
class IoUbalancedSmoothL1Loss(nn.Module):
    """reference mmdet/models/losses/iou_balanced_smooth_l1_loss.py:8-20."""

    def __init__(self, beta=1.0, delta=1.0, loss_weight=1.0):
        super(IoUbalancedSmoothL1Loss, self).__init__()
        self.beta, self.delta, self.loss_weight = beta, delta, loss_weight

    def forward_level(self, bbox_pred, bbox_targets, bbox_weights, iou, num_anchors, avg_factor):
        total = ops.smooth_l1_balanced_sum(bbox_pred, bbox_targets, bbox_weights, iou, num_anchors,
                                           self.beta, self.delta)
        return total * (self.loss_weight / avg_factor)

    def forward(self, pred, target, iou, weight, avg_factor=None, **kwargs):
        """pred / target / weight (N, 4); iou (N,)."""
        if pred.size() != target.size() or target.numel() == 0:
            raise AssertionError('pred / target shape mismatch or empty')
        if avg_factor is None:
            avg_factor = float((weight > 0).sum().item()) / 4 + 1e-6
        n = pred.shape[0]
        total = ops.smooth_l1_balanced_sum(pred.reshape(n, 4, 1, 1), target.reshape(n, 1, 4),
                                           weight.reshape(n, 1, 4), iou, 1, self.beta, self.delta)
        return total * (self.loss_weight / avg_factor)
