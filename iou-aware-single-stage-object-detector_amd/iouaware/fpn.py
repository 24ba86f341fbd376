"""Feature pyramid neck (reference mmdet/models/necks/fpn.py:9-136).

For RetinaNet: start_level=1, add_extra_convs=True, num_outs=5 -> laterals on
C3..C5, nearest x2 top-down sums, 3x3 output convs, P6 = 3x3/2 conv on C5,
P7 = 3x3/2 conv on P6 (no ReLU in between unless relu_before_extra_convs).
Parameters: lateral_convs.N.conv.*, fpn_convs.N.conv.*  Dense convs -> MIOpen.
"""
import torch.nn as nn
import torch.nn.functional as F

from .layers import ConvModule, xavier_init
from .registry import NECKS


@NECKS.register_module
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1,
                 add_extra_convs=False, extra_convs_on_inputs=True, relu_before_extra_convs=False,
                 conv_cfg=None, norm_cfg=None, activation=None):
        super(FPN, self).__init__()
        if not isinstance(in_channels, list):
            raise AssertionError('in_channels must be a list')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.activation = activation
        self.relu_before_extra_convs = relu_before_extra_convs
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            if num_outs < self.num_ins - start_level:
                raise AssertionError('num_outs too small')
        else:
            self.backbone_end_level = end_level
            if end_level > len(in_channels) or num_outs != end_level - start_level:
                raise AssertionError('inconsistent end_level / num_outs')
        self.start_level, self.end_level = start_level, end_level
        self.add_extra_convs, self.extra_convs_on_inputs = add_extra_convs, extra_convs_on_inputs

        common = dict(conv_cfg=conv_cfg, norm_cfg=norm_cfg, activation=activation, inplace=False)
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, **common))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, **common))
        extra = num_outs - self.backbone_end_level + start_level
        if add_extra_convs and extra >= 1:
            for i in range(extra):
                src = in_channels[self.backbone_end_level - 1] \
                    if (i == 0 and extra_convs_on_inputs) else out_channels
                self.fpn_convs.append(ConvModule(src, out_channels, 3, stride=2, padding=1,
                                                 **common))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')

    def forward(self, inputs):
        if len(inputs) != len(self.in_channels):
            raise AssertionError('FPN expects %d inputs' % len(self.in_channels))
        lat = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        n = len(lat)
        for i in range(n - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], scale_factor=2, mode='nearest')
        outs = [self.fpn_convs[i](lat[i]) for i in range(n)]
        if self.num_outs > n:
            if not self.add_extra_convs:
                for _ in range(self.num_outs - n):
                    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            else:
                first = inputs[self.backbone_end_level - 1] if self.extra_convs_on_inputs \
                    else outs[-1]
                outs.append(self.fpn_convs[n](first))
                for i in range(n + 1, self.num_outs):
                    src = F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]
                    outs.append(self.fpn_convs[i](src))
        return tuple(outs)
