"""`import mmdet...` aliases so code written against the reference's package
layout finds this build's implementations:

    from iouaware import compat; compat.install()
    from mmdet.models import build_detector          # -> iouaware.registry
    from mmdet.ops.nms import nms                     # -> HIP NMS
    from mmdet.ops import sigmoid_focal_loss          # -> HIP focal-loss op
    from mmdet.core import bbox2result, multiclass_nms, delta2bbox ...

Only the names on the IoU-aware RetinaNet path exist;
everything else of mmdetection is deliberately absent.
"""
import sys
import types


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def install(force=False):
    if 'mmdet' in sys.modules and not force:
        existing = sys.modules['mmdet']
        if getattr(existing, '__iouaware__', False):
            return existing
        raise RuntimeError('a different `mmdet` is already imported')
    from . import (anchors, api, bbox, detectors, fpn, head, layers, losses, nms_op, registry,
                   targets, backbones, dist as idist, focal_op, preprocess)
    root = _mod('mmdet', __version__='0.6.0+iouaware', __iouaware__=True)
    models = _mod('mmdet.models', **{k: getattr(registry, k) for k in (
        'BACKBONES', 'NECKS', 'ROI_EXTRACTORS', 'SHARED_HEADS', 'HEADS', 'LOSSES', 'DETECTORS',
        'build_backbone', 'build_neck', 'build_roi_extractor', 'build_shared_head', 'build_head',
        'build_loss', 'build_detector')})
    models.registry = _mod('mmdet.models.registry', Registry=registry.Registry, **{
        k: getattr(registry, k) for k in ('BACKBONES', 'NECKS', 'ROI_EXTRACTORS', 'SHARED_HEADS',
                                          'HEADS', 'LOSSES', 'DETECTORS')})
    models.builder = _mod('mmdet.models.builder', build=registry.build, **{
        k: getattr(registry, k) for k in ('build_backbone', 'build_neck', 'build_head',
                                          'build_loss', 'build_detector')})
    models.backbones = _mod('mmdet.models.backbones', ResNet=backbones.ResNet,
                            ResNeXt=backbones.ResNeXt, make_res_layer=backbones.make_res_layer)
    models.necks = _mod('mmdet.models.necks', FPN=fpn.FPN)
    models.anchor_heads = _mod('mmdet.models.anchor_heads', AnchorHead=head.AnchorHead,
                               IoUawareRetinaHead=head.IoUawareRetinaHead)
    models.detectors = _mod('mmdet.models.detectors', BaseDetector=detectors.BaseDetector,
                            SingleStageDetector=detectors.SingleStageDetector,
                            RetinaNet=detectors.RetinaNet)
    models.losses = _mod('mmdet.models.losses', FocalLoss=losses.FocalLoss,
                         SmoothL1Loss=losses.SmoothL1Loss,
                         IOUbalancedSigmoidFocalLoss=losses.IOUbalancedSigmoidFocalLoss,
                         IoUbalancedSmoothL1Loss=losses.IoUbalancedSmoothL1Loss)
    models.utils = _mod('mmdet.models.utils', **{k: getattr(layers, k) for k in (
        'ConvModule', 'build_conv_layer', 'build_norm_layer', 'xavier_init', 'normal_init',
        'uniform_init', 'kaiming_init', 'bias_init_with_prob')})
    for k in ('ResNet', 'ResNeXt'):
        setattr(models, k, getattr(backbones, k))
    models.FPN, models.IoUawareRetinaHead, models.RetinaNet = fpn.FPN, head.IoUawareRetinaHead, \
        detectors.RetinaNet
    models.SingleStageDetector = detectors.SingleStageDetector

    ops_nms = _mod('mmdet.ops.nms', nms=nms_op.nms, soft_nms=nms_op.soft_nms)
    ops_nms.nms_wrapper = _mod('mmdet.ops.nms.nms_wrapper', nms=nms_op.nms, soft_nms=nms_op.soft_nms)
    ops_fl = _mod('mmdet.ops.sigmoid_focal_loss', SigmoidFocalLoss=focal_op.SigmoidFocalLoss,
                  sigmoid_focal_loss=focal_op.sigmoid_focal_loss)
    mops = _mod('mmdet.ops', nms=nms_op.nms, soft_nms=nms_op.soft_nms,
                SigmoidFocalLoss=focal_op.SigmoidFocalLoss,
                sigmoid_focal_loss=focal_op.sigmoid_focal_loss)
    mops.nms_module, mops.sigmoid_focal_loss_module = ops_nms, ops_fl

    core = _mod('mmdet.core', AnchorGenerator=anchors.AnchorGenerator,
                anchor_target=targets.anchor_target, delta2bbox=bbox.delta2bbox,
                bbox2delta=bbox.bbox2delta, bbox_overlaps=bbox.bbox_overlaps,
                bbox2result=bbox.bbox2result, multi_apply=bbox.multi_apply,
                multiclass_nms=nms_op.multiclass_nms, MaxIoUAssigner=targets.MaxIoUAssigner,
                PseudoSampler=targets.PseudoSampler, build_assigner=targets.build_assigner)
    core.anchor = _mod('mmdet.core.anchor', AnchorGenerator=anchors.AnchorGenerator,
                       anchor_target=targets.anchor_target)
    core.bbox = _mod('mmdet.core.bbox', delta2bbox=bbox.delta2bbox, bbox2delta=bbox.bbox2delta,
                     bbox_overlaps=bbox.bbox_overlaps, bbox2result=bbox.bbox2result,
                     MaxIoUAssigner=targets.MaxIoUAssigner, PseudoSampler=targets.PseudoSampler,
                     build_assigner=targets.build_assigner)
    core.post_processing = _mod('mmdet.core.post_processing', multiclass_nms=nms_op.multiclass_nms)

    apis = _mod('mmdet.apis', init_dist=idist.init_dist, init_detector=api.init_detector,
                get_dist_info=idist.get_dist_info)
    datasets = _mod('mmdet.datasets', ImageTransform=preprocess.ImageTransform)
    datasets.transforms = _mod('mmdet.datasets.transforms', ImageTransform=preprocess.ImageTransform)
    root.models, root.ops, root.core, root.apis, root.datasets = models, mops, core, apis, datasets
    return root
