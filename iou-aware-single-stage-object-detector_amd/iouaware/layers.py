"""Conv / norm / init building blocks used by the backbone, neck and head
(reference mmdet/models/utils/{conv_module,norm,weight_init}.py and the mmcv
0.2.8 init helpers of the same names).  The convolutions themselves are dense
contractions and stay on PyTorch-ROCm (MIOpen / rocBLAS on MFMA)."""
import math
import warnings

import torch.nn as nn


# ---- weight init (mmcv.cnn.{constant,normal,uniform,xavier,kaiming}_init semantics)
def _set_bias(module, bias):
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    nn.init.constant_(module.weight, val)
    _set_bias(module, bias)


def normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    _set_bias(module, bias)


def uniform_init(module, a=0, b=1, bias=0):
    nn.init.uniform_(module.weight, a, b)
    _set_bias(module, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if distribution not in ('uniform', 'normal'):
        raise AssertionError(distribution)
    init = nn.init.xavier_uniform_ if distribution == 'uniform' else nn.init.xavier_normal_
    init(module.weight, gain=gain)
    _set_bias(module, bias)


def kaiming_init(module, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    if distribution not in ('uniform', 'normal'):
        raise AssertionError(distribution)
    init = nn.init.kaiming_uniform_ if distribution == 'uniform' else nn.init.kaiming_normal_
    init(module.weight, mode=mode, nonlinearity=nonlinearity)
    _set_bias(module, bias)


def bias_init_with_prob(prior_prob):
    """bias b with sigmoid(b) == prior_prob (retina_cls: -log(99) for 0.01)"""
    return float(-math.log((1 - prior_prob) / prior_prob))


# ---- layer factories
_CONV_TYPES = {'Conv': nn.Conv2d}
_NORM_TYPES = {'BN': ('bn', nn.BatchNorm2d), 'SyncBN': ('bn', nn.SyncBatchNorm),
               'GN': ('gn', nn.GroupNorm)}


def build_conv_layer(cfg, *args, **kwargs):
    spec = dict(type='Conv') if cfg is None else dict(cfg)
    if 'type' not in spec:
        raise AssertionError('conv cfg needs a "type"')
    kind = spec.pop('type')
    if kind not in _CONV_TYPES:
        raise KeyError('Unrecognized conv type {}'.format(kind))
    return _CONV_TYPES[kind](*args, **kwargs, **spec)


def build_norm_layer(cfg, num_features, postfix=''):
    """-> (attribute name, layer); name = abbreviation + postfix ('bn1', 'gn', ...)."""
    if not (isinstance(cfg, dict) and 'type' in cfg):
        raise AssertionError('norm cfg needs a "type"')
    spec = dict(cfg)
    kind = spec.pop('type')
    if kind not in _NORM_TYPES:
        raise KeyError('Unrecognized norm type {}'.format(kind))
    abbr, cls = _NORM_TYPES[kind]
    if not isinstance(postfix, (int, str)):
        raise AssertionError('postfix must be int or str')
    trainable = spec.pop('requires_grad', True)
    spec.setdefault('eps', 1e-5)
    if kind == 'GN':
        if 'num_groups' not in spec:
            raise AssertionError('GN needs num_groups')
        layer = cls(num_channels=num_features, **spec)
    else:
        layer = cls(num_features, **spec)
    for p in layer.parameters():
        p.requires_grad = trainable
    return abbr + str(postfix), layer


class ConvModule(nn.Module):
    """conv -> [norm] -> [relu]; parameters live under `.conv` (and `.bn`/`.gn`),
    which is what reference checkpoints name them."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None, activation='relu',
                 inplace=True, activate_last=True):
        super(ConvModule, self).__init__()
        if not (conv_cfg is None or isinstance(conv_cfg, dict)):
            raise AssertionError('conv_cfg must be None or dict')
        if not (norm_cfg is None or isinstance(norm_cfg, dict)):
            raise AssertionError('norm_cfg must be None or dict')
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        self.activation, self.inplace, self.activate_last = activation, inplace, activate_last
        self.with_norm = norm_cfg is not None
        self.with_activatation = activation is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.with_bias = bias
        if self.with_norm and self.with_bias:
            warnings.warn('ConvModule has norm and bias at the same time')
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size,
                                     stride=stride, padding=padding, dilation=dilation,
                                     groups=groups, bias=bias)
        for attr in ('in_channels', 'out_channels', 'kernel_size', 'stride', 'padding',
                     'dilation', 'transposed', 'output_padding', 'groups'):
            setattr(self, attr, getattr(self.conv, attr))
        if self.with_norm:
            width = out_channels if activate_last else in_channels
            self.norm_name, norm = build_norm_layer(norm_cfg, width)
            self.add_module(self.norm_name, norm)
        if self.with_activatation:
            if activation != 'relu':
                raise ValueError('{} is currently not supported.'.format(activation))
            self.activate = nn.ReLU(inplace=inplace)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name)

    def init_weights(self):
        kaiming_init(self.conv, nonlinearity='relu' if self.activation is None else self.activation)
        if self.with_norm:
            constant_init(self.norm, 1, bias=0)

    def forward(self, x, activate=True, norm=True):
        do_norm = norm and self.with_norm
        do_act = activate and self.with_activatation
        if self.activate_last:
            x = self.conv(x)
            if do_norm:
                x = self.norm(x)
            return self.activate(x) if do_act else x
        if do_norm:
            x = self.norm(x)
        if do_act:
            x = self.activate(x)
        return self.conv(x)
