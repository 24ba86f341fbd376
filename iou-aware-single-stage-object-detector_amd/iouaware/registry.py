"""Name -> class registries and the cfg-dict builder: the drop-in boundary of
the reference's model zoo (reference mmdet/models/registry.py:4-45,
mmdet/models/builder.py:8-60).  `type=` strings of the configs resolve here, so
configs/iou_aware_single_stage_detector/*.py load unchanged.

Behaviour kept: only nn.Module subclasses register (TypeError otherwise);
registering a name twice is a KeyError; building an unknown `type` string is a
KeyError naming the registry; a non-str / non-class `type` is a TypeError;
remaining keys become constructor kwargs; default_args fill missing keys only;
a list of cfgs builds an nn.Sequential.
"""
import torch.nn as nn


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._classes = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._classes

    def get(self, key):
        return self._classes.get(key)

    def __contains__(self, key):
        return key in self._classes

    def __len__(self):
        return len(self._classes)

    def __repr__(self):
        return 'Registry(name=%s, items=%s)' % (self._name, sorted(self._classes))

    def register_module(self, cls):
        """class decorator"""
        if not (isinstance(cls, type) and issubclass(cls, nn.Module)):
            raise TypeError('module must be a child of nn.Module, but got {}'.format(cls))
        key = cls.__name__
        if key in self._classes:
            raise KeyError('{} is already registered in {}'.format(key, self._name))
        self._classes[key] = cls
        return cls


BACKBONES = Registry('backbone')
NECKS = Registry('neck')
ROI_EXTRACTORS = Registry('roi_extractor')
SHARED_HEADS = Registry('shared_head')
HEADS = Registry('head')
LOSSES = Registry('loss')
DETECTORS = Registry('detector')


def _resolve(cfg, registry, default_args):
    if not (isinstance(cfg, dict) and 'type' in cfg):
        raise AssertionError('cfg must be a dict with a "type" key, got %r' % (cfg,))
    if not (default_args is None or isinstance(default_args, dict)):
        raise AssertionError('default_args must be a dict or None')
    kwargs = dict(cfg)
    kind = kwargs.pop('type')
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError('{} is not in the {} registry'.format(kind, registry.name))
    elif isinstance(kind, type):
        cls = kind
    else:
        raise TypeError('type must be a str or valid type, but got {}'.format(type(kind)))
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    return cls(**kwargs)


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[_resolve(c, registry, default_args) for c in cfg])
    return _resolve(cfg, registry, default_args)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_roi_extractor(cfg):
    return build(cfg, ROI_EXTRACTORS)


def build_shared_head(cfg):
    return build(cfg, SHARED_HEADS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
