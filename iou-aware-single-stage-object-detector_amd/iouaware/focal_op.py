"""`mmdet.ops.sigmoid_focal_loss` / `SigmoidFocalLoss` with the reference op's
semantics (integer targets 0..C, no per-anchor weight, elementwise loss then
mean / sum in Python; reference mmdet/ops/sigmoid_focal_loss/functions/
sigmoid_focal_loss.py:8-42, modules/sigmoid_focal_loss.py:6-23)."""
import torch.nn as nn

from . import ops


def sigmoid_focal_loss(input, target, gamma=2.0, alpha=0.25, reduction='mean'):
    loss = ops.sigmoid_focal_loss_elementwise(input, target, gamma, alpha)
    if reduction == 'none':
        return loss
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'sum':
        return loss.sum()
    raise ValueError('{} is not a valid value for reduction'.format(reduction))


class SigmoidFocalLoss(nn.Module):
    def __init__(self, gamma, alpha):
        super(SigmoidFocalLoss, self).__init__()
        self.gamma, self.alpha = gamma, alpha

    def forward(self, logits, targets):
        if not logits.is_cuda:
            raise AssertionError('logits must be on a ROCm device')
        return sigmoid_focal_loss(logits, targets, self.gamma, self.alpha).sum()

    def __repr__(self):
        return '{}(gamma={}, alpha={})'.format(self.__class__.__name__, self.gamma, self.alpha)
