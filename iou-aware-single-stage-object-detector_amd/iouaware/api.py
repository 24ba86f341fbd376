"""High-level helpers: build a detector from a reference config file and run
batched inference on already pre-processed tensors (reference
mmdet/apis/inference.py:15-94 `init_detector`; its `inference_detector` image
pipeline needs mmcv image ops and is outside the hot path)."""
import torch

from .checkpoint import load_checkpoint
from .config import Config
from .registry import build_detector


def init_detector(config, checkpoint=None, device='cuda:0'):
    if isinstance(config, str):
        config = Config.fromfile(config)
    elif not isinstance(config, Config):
        raise TypeError('config must be a filename or Config object, but got {}'.format(
            type(config)))
    config.model.pretrained = None
    model = build_detector(config.model, test_cfg=config.test_cfg)
    if checkpoint is not None:
        ckpt = load_checkpoint(model, checkpoint)
        meta = ckpt.get('meta', {}) if isinstance(ckpt, dict) else {}
        if 'CLASSES' in meta:
            model.CLASSES = meta['CLASSES']
    model.cfg = config
    model.to(device)
    model.eval()
    return model


@torch.no_grad()
def inference_batch(model, imgs, img_metas, rescale=True):
    """imgs (B,3,H,W) normalised + padded tensor -> list over images of per-class ndarrays."""
    return model.simple_test_batch(imgs, img_metas, rescale=rescale)
