"""ResNet / ResNeXt feature extractors (C2..C5) on PyTorch-ROCm.

Host-side modules only: the 3x3 / 1x1 / grouped convolutions are dense
contractions that MIOpen / rocBLAS execute on MFMA; nothing here is a
hand-written kernel.  Constructor kwargs, parameter names (`conv1`, `bn1`,
`layerN.M.conv{1,2,3}`, `layerN.M.bn{1,2,3}`, `layerN.M.downsample.{0,1}`) and
train()/init semantics follow the reference so its configs and checkpoints
drop in (reference mmdet/models/backbones/resnet.py:88-267,333-527;
resnext.py:12-91,157-226).  Deformable conv, GCNet and attention plugins are
outside the IoU-aware RetinaNet path and are rejected explicitly.
"""
import math

import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from .layers import build_conv_layer, build_norm_layer, constant_init, kaiming_init
from .registry import BACKBONES


def _reject_plugins(dcn, gcb, gen_attention):
    if dcn is not None or gcb is not None or gen_attention is not None:
        raise NotImplementedError('dcn / gcb / gen_attention are outside the IoU-aware '
                                  'RetinaNet hot path of this build')


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch',
                 with_cp=False, conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, gcb=None,
                 gen_attention=None, groups=1, base_width=4):
        super(BasicBlock, self).__init__()
        _reject_plugins(dcn, gcb, gen_attention)
        if with_cp:
            raise AssertionError('with_cp is not supported by BasicBlock')
        self.norm1_name, n1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, n2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride,
                                      padding=dilation, dilation=dilation, bias=False)
        self.add_module(self.norm1_name, n1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.add_module(self.norm2_name, n2)
        self.relu = nn.ReLU(inplace=True)
        self.downsample, self.stride, self.dilation = downsample, stride, dilation

    norm1 = property(lambda self: getattr(self, self.norm1_name))
    norm2 = property(lambda self: getattr(self, self.norm2_name))

    def forward(self, x):
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        out = out + (x if self.downsample is None else self.downsample(x))
        return self.relu(out)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (stride here for style='pytorch', grouped for ResNeXt) -> 1x1."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch',
                 with_cp=False, conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, gcb=None,
                 gen_attention=None, groups=1, base_width=4):
        super(Bottleneck, self).__init__()
        if style not in ('pytorch', 'caffe'):
            raise AssertionError(style)
        _reject_plugins(dcn, gcb, gen_attention)
        self.inplanes, self.planes, self.stride, self.dilation = inplanes, planes, stride, dilation
        self.style, self.with_cp, self.conv_cfg, self.norm_cfg = style, with_cp, conv_cfg, norm_cfg
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
        self.conv1_stride, self.conv2_stride = s1, s2
        width = planes if groups == 1 else int(math.floor(planes * (base_width / 64))) * groups
        self.norm1_name, n1 = build_norm_layer(norm_cfg, width, postfix=1)
        self.norm2_name, n2 = build_norm_layer(norm_cfg, width, postfix=2)
        self.norm3_name, n3 = build_norm_layer(norm_cfg, planes * self.expansion, postfix=3)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, width, kernel_size=1, stride=s1,
                                      bias=False)
        self.add_module(self.norm1_name, n1)
        self.conv2 = build_conv_layer(conv_cfg, width, width, kernel_size=3, stride=s2,
                                      padding=dilation, dilation=dilation, groups=groups,
                                      bias=False)
        self.add_module(self.norm2_name, n2)
        self.conv3 = build_conv_layer(conv_cfg, width, planes * self.expansion, kernel_size=1,
                                      bias=False)
        self.add_module(self.norm3_name, n3)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    norm1 = property(lambda self: getattr(self, self.norm1_name))
    norm2 = property(lambda self: getattr(self, self.norm2_name))
    norm3 = property(lambda self: getattr(self, self.norm3_name))

    def _residual(self, x):
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.relu(self.norm2(self.conv2(out)))
        out = self.norm3(self.conv3(out))
        return out + (x if self.downsample is None else self.downsample(x))

    def forward(self, x):
        if self.with_cp and x.requires_grad:
            import torch.utils.checkpoint as cp
            return self.relu(cp.checkpoint(self._residual, x))
        return self.relu(self._residual(x))


def make_res_layer(block, inplanes, planes, blocks, stride=1, dilation=1, style='pytorch',
                   with_cp=False, conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, gcb=None,
                   gen_attention=None, gen_attention_blocks=(), groups=1, base_width=4):
    out_ch = planes * block.expansion
    downsample = None
    if stride != 1 or inplanes != out_ch:
        downsample = nn.Sequential(
            build_conv_layer(conv_cfg, inplanes, out_ch, kernel_size=1, stride=stride, bias=False),
            build_norm_layer(norm_cfg, out_ch)[1])
    common = dict(style=style, with_cp=with_cp, conv_cfg=conv_cfg, norm_cfg=norm_cfg, dcn=dcn,
                  gcb=gcb, groups=groups, base_width=base_width)
    layers = [block(inplanes, planes, stride, dilation, downsample,
                    gen_attention=gen_attention if 0 in gen_attention_blocks else None, **common)]
    for i in range(1, blocks):
        layers.append(block(out_ch, planes, 1, dilation,
                            gen_attention=gen_attention if i in gen_attention_blocks else None,
                            **common))
    return nn.Sequential(*layers)


@BACKBONES.register_module
class ResNet(nn.Module):
    """depth in {18,34,50,101,152}; returns the stage outputs listed in out_indices."""

    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)),
                     50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)),
                     152: (Bottleneck, (3, 8, 36, 3))}
    _groups, _base_width = 1, 4

    def __init__(self, depth, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), style='pytorch', frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), gcb=None,
                 stage_with_gcb=(False, False, False, False), gen_attention=None,
                 stage_with_gen_attention=((), (), (), ()), with_cp=False,
                 zero_init_residual=True):
        super(ResNet, self).__init__()
        if depth not in self.arch_settings:
            raise KeyError('invalid depth {} for resnet'.format(depth))
        if not 1 <= num_stages <= 4:
            raise AssertionError('num_stages must be in 1..4')
        if not len(strides) == len(dilations) == num_stages:
            raise AssertionError('strides / dilations must have num_stages entries')
        if max(out_indices) >= num_stages:
            raise AssertionError('out_indices exceed num_stages')
        _reject_plugins(dcn, gcb, gen_attention)
        self.depth, self.num_stages, self.strides, self.dilations = depth, num_stages, strides, dilations
        self.out_indices, self.style, self.frozen_stages = out_indices, style, frozen_stages
        self.conv_cfg, self.norm_cfg, self.with_cp, self.norm_eval = conv_cfg, norm_cfg, with_cp, norm_eval
        self.dcn, self.gcb, self.gen_attention = dcn, gcb, gen_attention
        self.stage_with_dcn, self.stage_with_gcb = stage_with_dcn, stage_with_gcb
        self.zero_init_residual = zero_init_residual
        self.block, stage_blocks = self.arch_settings[depth]
        self.stage_blocks = stage_blocks[:num_stages]

        # stem
        self.conv1 = build_conv_layer(conv_cfg, 3, 64, kernel_size=7, stride=2, padding=3,
                                      bias=False)
        self.norm1_name, n1 = build_norm_layer(norm_cfg, 64, postfix=1)
        self.add_module(self.norm1_name, n1)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

        self.inplanes = 64
        self.res_layers = []
        for i, nblocks in enumerate(self.stage_blocks):
            planes = 64 * 2 ** i
            layer = make_res_layer(self.block, self.inplanes, planes, nblocks, stride=strides[i],
                                   dilation=dilations[i], style=style, with_cp=with_cp,
                                   conv_cfg=conv_cfg, norm_cfg=norm_cfg, groups=self._groups,
                                   base_width=self._base_width)
            self.inplanes = planes * self.block.expansion
            name = 'layer{}'.format(i + 1)
            self.add_module(name, layer)
            self.res_layers.append(name)
        self._freeze_stages()
        self.feat_dim = self.block.expansion * 64 * 2 ** (len(self.stage_blocks) - 1)

    norm1 = property(lambda self: getattr(self, self.norm1_name))

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.norm1.eval()
            for m in (self.conv1, self.norm1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            stage = getattr(self, 'layer{}'.format(i))
            stage.eval()
            for p in stage.parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
            return
        if pretrained is not None:
            raise TypeError('pretrained must be a str or None')
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                kaiming_init(m)
            elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
                constant_init(m, 1)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    constant_init(m.norm3, 0)
                elif isinstance(m, BasicBlock):
                    constant_init(m.norm2, 0)

    def forward(self, x):
        x = self.maxpool(self.relu(self.norm1(self.conv1(x))))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super(ResNet, self).train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self


@BACKBONES.register_module
class ResNeXt(ResNet):
    """ResNet with grouped 3x3 convolutions (groups x base_width), e.g. 32x4d / 64x4d."""

    arch_settings = {50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)),
                     152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, groups=1, base_width=4, **kwargs):
        self._groups, self._base_width = groups, base_width
        super(ResNeXt, self).__init__(**kwargs)
        self.groups, self.base_width = groups, base_width
