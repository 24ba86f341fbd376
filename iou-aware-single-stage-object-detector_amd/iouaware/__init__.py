"""MI355X-native IoU-aware RetinaNet hot path (host side).

Importing this package registers the reference's type names (`RetinaNet`,
`ResNet`, `ResNeXt`, `FPN`, `IoUawareRetinaHead`, `FocalLoss`, `SmoothL1Loss`)
so `configs/iou_aware_single_stage_detector/*.py` of the reference build
unchanged through `build_detector`.  The post-conv compute lives in
libiouaware_hip.so (hand-written gfx950 kernels, C-ABI in include/iouaware.h).
"""
from .config import Config, ConfigDict                                  # noqa: F401
from .registry import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, Registry,  # noqa: F401
                       build_backbone, build_detector, build_head, build_loss, build_neck)
from . import backbones, fpn, losses, head, detectors                   # noqa: F401  (register)
from .api import init_detector, inference_batch                         # noqa: F401
from .preprocess import ImageTransform                                  # noqa: F401

__version__ = '0.1.0'
