"""Training-step glue for BASELINE config 5 (R-50 training step with the HIP loss kernels,
data-parallel): the pieces of the reference's training driver that touch the hot path
(reference mmdet/apis/train.py:18-45 `parse_losses` / `batch_processor`,
mmdet/core/utils/dist_utils.py:9-57 `allreduce_grads` / `DistOptimizerHook`).

The mmcv Runner / hook machinery is out of scope; `train_step` is the body of one iteration:
forward_train -> parse_losses -> zero_grad -> backward -> (gradient all-reduce) -> clip -> step.
Gradient averaging over ranks is torch DistributedDataParallel (bucketed all-reduce on RCCL,
overlapped with backward) instead of the reference's single flat all-reduce after backward.
"""
from collections import OrderedDict

import torch
import torch.distributed as dist
from torch.nn.utils import clip_grad_norm_


def parse_losses(losses):
    """dict of tensors / lists of tensors -> (total loss, log_vars).  A key contributes to the
    total iff its name contains 'loss' -- which includes the head's 'losses_iou' key."""
    log_vars = OrderedDict()
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, list):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError('{} is not a tensor or list of tensors'.format(name))
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    return loss, log_vars


def build_optimizer(model, optimizer_cfg):
    """`optimizer = dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0001)` of the configs."""
    cfg = dict(optimizer_cfg)
    kind = cfg.pop('type')
    params = [p for p in model.parameters() if p.requires_grad]
    return getattr(torch.optim, kind)(params, **cfg)


def wrap_ddp(model, device_ids=None):
    """DistributedDataParallel over the current process group (RCCL for backend 'nccl')."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    from torch.nn.parallel import DistributedDataParallel
    return DistributedDataParallel(model, device_ids=device_ids, broadcast_buffers=False)


def train_step(model, optimizer, img, img_meta, gt_bboxes, gt_labels, grad_clip=None):
    """One iteration.  grad_clip: dict(max_norm=35, norm_type=2) like optimizer_config.grad_clip.
    Returns the log_vars of parse_losses as python floats (one host sync at the end)."""
    losses = model(img, img_meta, return_loss=True, gt_bboxes=gt_bboxes, gt_labels=gt_labels)
    if losses is None:                       # an image without valid anchors (reference :362-363)
        return None
    loss, log_vars = parse_losses(losses)
    optimizer.zero_grad()
    loss.backward()
    if grad_clip is not None:
        clip_grad_norm_([p for p in model.parameters() if p.requires_grad and p.grad is not None],
                        **grad_clip)
    optimizer.step()
    return OrderedDict((k, float(v.detach())) for k, v in log_vars.items())
