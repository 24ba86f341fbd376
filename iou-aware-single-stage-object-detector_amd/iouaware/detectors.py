"""Detector glue: BaseDetector.forward dispatch, SingleStageDetector =
backbone -> neck -> head, RetinaNet thin subclass (reference
mmdet/models/detectors/{base,single_stage,retinanet}.py).

Kept: registry names, ctor kwargs, `forward(img, img_meta, return_loss=True,
**kwargs)`, the fork's extra positional `gt_bboxes, gt_labels` at test time
(base.py:62-67), `rescale`, `simple_test` returning image 0's per-class
ndarray list for a batch of one.  Lifted: the batch-1 assert of the reference
(base.py:96-98) -- `simple_test_batch` / `forward_test` handle B images per
call (BASELINE configs 2, 3) and the whole post-conv path is one library call.
"""
import logging

import torch
import torch.nn as nn

from . import registry
from .bbox import bbox2result
from .registry import DETECTORS


class BaseDetector(nn.Module):
    def __init__(self):
        super(BaseDetector, self).__init__()

    with_neck = property(lambda self: getattr(self, 'neck', None) is not None)
    with_bbox = property(lambda self: getattr(self, 'bbox_head', None) is not None)

    def init_weights(self, pretrained=None):
        if pretrained is not None:
            logging.getLogger().info('load model from: {}'.format(pretrained))

    def extract_feats(self, imgs):
        if not isinstance(imgs, list):
            raise AssertionError('imgs must be a list')
        for img in imgs:
            yield self.extract_feat(img)

    def forward_test(self, imgs, img_metas, gt_bboxes=None, gt_labels=None, **kwargs):
        for var, name in ((imgs, 'imgs'), (img_metas, 'img_metas')):
            if not isinstance(var, list):
                raise TypeError('{} must be a list, but got {}'.format(name, type(var)))
        if len(imgs) != len(img_metas):
            raise ValueError('num of augmentations ({}) != num of image meta ({})'.format(
                len(imgs), len(img_metas)))
        if len(imgs) != 1:
            return self.aug_test(imgs, img_metas, **kwargs)
        gtb = gt_bboxes[0] if gt_bboxes is not None else None
        gtl = gt_labels[0] if gt_labels is not None else None
        return self.simple_test(imgs[0], img_metas[0], gtb, gtl, **kwargs)

    def forward(self, img, img_meta, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_meta, **kwargs)
        return self.forward_test(img, img_meta, **kwargs)


@DETECTORS.register_module
class SingleStageDetector(BaseDetector):
    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None):
        super(SingleStageDetector, self).__init__()
        self.backbone = registry.build_backbone(backbone)
        if neck is not None:
            self.neck = registry.build_neck(neck)
        self.bbox_head = registry.build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_weights(pretrained=pretrained)

    def init_weights(self, pretrained=None):
        super(SingleStageDetector, self).init_weights(pretrained)
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            for m in (self.neck if isinstance(self.neck, nn.Sequential) else [self.neck]):
                m.init_weights()
        self.bbox_head.init_weights()

    def extract_feat(self, img):
        x = self.backbone(img)
        return self.neck(x) if self.with_neck else x

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        outs = self.bbox_head(self.extract_feat(img))
        return self.bbox_head.loss(*(outs + (gt_bboxes, gt_labels, img_metas, self.train_cfg)),
                                   gt_bboxes_ignore=gt_bboxes_ignore)

    def forward_head(self, img):
        """backbone + neck + head convolutions -> (cls[L], reg[L], iou[L])"""
        return self.bbox_head(self.extract_feat(img))

    def simple_test_device(self, img, img_meta, rescale=False):
        """whole batch, results left on the device: dets (B,max,5), labels, rows, num."""
        outs = self.forward_head(img)
        return self.bbox_head.get_bboxes_batched(*outs, img_meta, self.test_cfg, rescale)

    def simple_test_batch(self, img, img_meta, gt_bboxes=None, gt_labels=None, rescale=False):
        """-> list over images of per-class ndarray lists (bbox2result, reference
        core/bbox/transforms.py:148-166).  The whole batch comes back in ONE device-to-host copy
        of fixed-size records (the reference copies detections and labels of every image
        separately, two synchronisations per image)."""
        if hasattr(self.bbox_head, 'get_bboxes_batched') and img.is_cuda:
            from .dist import pack_detections, unpack_detections
            dets, labels, _, num = self.simple_test_device(img, img_meta, rescale)
            rec = pack_detections(dets, labels, num).cpu()          # the one host sync per batch
            dets, labels, num = unpack_detections(rec, dets.shape[1])
            dets, labels, num = dets.numpy(), labels.numpy(), num.tolist()
            return [bbox2result(dets[b, :k], labels[b, :k], self.bbox_head.num_classes)
                    for b, k in enumerate(num)]
        outs = self.forward_head(img)
        bbox_list = self.bbox_head.get_bboxes(*(outs + (gt_bboxes, gt_labels, img_meta,
                                                        self.test_cfg, rescale)))
        return [bbox2result(d, l, self.bbox_head.num_classes) for d, l in bbox_list]

    def simple_test_batch_submit(self, img, img_meta, rescale=False):
        """Asynchronous half of simple_test_batch for a serving loop: enqueues the network, the
        decode / NMS stage and the one device-to-host copy (into a pinned buffer) and returns
        at once; `PendingResults.collect()` waits for that copy and builds the per-class arrays.
        Submitting batch i+1 before collecting batch i overlaps the host's bbox2result with the
        device's work (the records of batch i are complete before batch i+1 touches any shared
        workspace: same stream)."""
        from .dist import pack_detections
        if not (hasattr(self.bbox_head, 'get_bboxes_batched') and img.is_cuda):
            raise RuntimeError('simple_test_batch_submit needs the batched HIP head on a ROCm device')
        dets, labels, _, num = self.simple_test_device(img, img_meta, rescale)
        rec = pack_detections(dets, labels, num)
        host = torch.empty(rec.shape, dtype=rec.dtype, pin_memory=True)
        host.copy_(rec, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return PendingResults(host, done, dets.shape[1], self.bbox_head.num_classes)

    def simple_test(self, img, img_meta, gt_bboxes=None, gt_labels=None, rescale=False):
        results = self.simple_test_batch(img, img_meta, gt_bboxes, gt_labels, rescale)
        return results[0] if len(results) == 1 else results

    def aug_test(self, imgs, img_metas, rescale=False):
        raise NotImplementedError


class PendingResults(object):
    """detections of one submitted batch, still on their way to the host"""

    def __init__(self, host, done, max_per_img, num_classes):
        self._host, self._done, self._m, self._nc = host, done, max_per_img, num_classes

    def collect(self):
        from .dist import unpack_detections
        self._done.synchronize()
        dets, labels, num = unpack_detections(self._host, self._m)
        dets, labels, num = dets.numpy(), labels.numpy(), num.tolist()
        return [bbox2result(dets[b, :k], labels[b, :k], self._nc) for b, k in enumerate(num)]


@DETECTORS.register_module
class RetinaNet(SingleStageDetector):
    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None):
        super(RetinaNet, self).__init__(backbone, neck, bbox_head, train_cfg, test_cfg, pretrained)
