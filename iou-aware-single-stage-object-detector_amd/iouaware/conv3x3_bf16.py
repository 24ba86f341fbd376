"""The conv towers and the class-output convolution of an IoUawareRetinaHead in bf16 on this
library's MFMA implicit-GEMM kernel (csrc/conv3x3_bf16.hip) -- BASELINE config 3 (R-101 bf16).

Reference: mmdet/models/anchor_heads/iou_aware_retina_head.py:171-219 (4 + 4 tower ConvModules
256 -> 256 with ReLU, retina_cls 256 -> A*C, retina_reg / retina_iou), weights shared by the five
pyramid levels.  Per tower layer ONE launch covers all levels and both towers (the towers are two
groups reading / writing the channel halves of one 2F-channel activation); bias and ReLU are the
kernel's epilogue; retina_reg | retina_iou (36 + 9 output channels) are one more launch on the reg
tower (padded to 46 channels, split afterwards).
"""
import collections

import torch

from . import ops


class Bf16ConvHead(object):
    def __init__(self, head):
        convs_c, convs_r = list(head.cls_convs), list(head.reg_convs)
        if any(m.with_norm or not m.with_activatation for m in convs_c + convs_r):
            raise NotImplementedError('towers with norm layers / without ReLU')
        if len(convs_c) != len(convs_r) or not convs_c:
            raise NotImplementedError('towers of different depth')
        from .fuse import _wino_ok               # plain 3x3 / stride 1 / pad 1 / dilation 1 / groups 1
        plain = [m.conv for m in convs_c + convs_r] + [head.retina_cls]
        if not all(_wino_ok(c) and getattr(c, 'padding_mode', 'zeros') == 'zeros' for c in plain):
            raise NotImplementedError('the kernel covers plain 3x3 / stride-1 / pad-1 convolutions')
        F = head.feat_channels
        if head.in_channels % 32 or F % 32 or head.retina_cls.out_channels % 2:
            raise NotImplementedError('channel counts')
        self.F, self.cin, self.n_layers = F, head.in_channels, len(convs_c)
        self.head = head

        def wb(m):
            conv = m.conv if hasattr(m, 'conv') else m
            b = conv.bias if conv.bias is not None else torch.zeros(conv.out_channels, device=conv.weight.device)
            return conv.weight.detach(), b.detach().float()
        # layer 0: both towers read the FPN feature -> one group with 2F output channels
        (wc, bc), (wr, br) = wb(convs_c[0]), wb(convs_r[0])
        self.w0 = ops.conv3x3_bf16_pack(torch.cat([wc, wr], 0))
        self.b0 = torch.cat([bc, br]).contiguous()
        # layers 1..: two groups (cls tower = channels [0, F), reg tower = [F, 2F))
        self.w, self.b = [], []
        for i in range(1, self.n_layers):
            (wc, bc), (wr, br) = wb(convs_c[i]), wb(convs_r[i])
            self.w.append(ops.conv3x3_bf16_pack(torch.cat([wc, wr], 0), groups=2))
            self.b.append(torch.cat([bc, br]).contiguous())
        wcls, self.b_cls = wb(head.retina_cls)
        self.w_cls = ops.conv3x3_bf16_pack(wcls)
        self.c_cls = head.retina_cls.out_channels
        # retina_reg | retina_iou (36 + 9 output channels) as ONE convolution on the reg tower, padded
        # to an even channel count (the kernel stores channel pairs); the outputs are split afterwards
        def plain3(c):
            return (tuple(c.kernel_size) == (3, 3) and tuple(c.stride) == (1, 1) and tuple(c.padding) == (1, 1)
                    and tuple(c.dilation) == (1, 1) and c.groups == 1 and getattr(c, 'padding_mode', 'zeros') == 'zeros')
        if not (plain3(head.retina_reg) and plain3(head.retina_iou)):
            raise NotImplementedError('the kernel covers plain 3x3 / stride-1 / pad-1 convolutions')
        (wr, brg), (wi, bi) = wb(head.retina_reg), wb(head.retina_iou)
        self.c_reg, self.c_iou = head.retina_reg.out_channels, head.retina_iou.out_channels
        self.c_ri = (self.c_reg + self.c_iou + 1) // 2 * 2
        pad = self.c_ri - self.c_reg - self.c_iou
        w_ri = torch.cat([wr, wi] + ([torch.zeros((pad,) + tuple(wr.shape[1:]), dtype=wr.dtype, device=wr.device)] if pad else []), 0)
        self.w_ri = ops.conv3x3_bf16_pack(w_ri)
        self.b_ri = torch.cat([brg, bi] + ([torch.zeros(pad, device=brg.device)] if pad else [])).contiguous()
        self._bufs = collections.OrderedDict()

    def usable(self, feats):
        return (not torch.is_grad_enabled()) and all(
            x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] == self.cin
            and x.is_contiguous(memory_format=torch.channels_last) for x in feats)

    def _acts(self, name, feats, channels):
        key = (name, channels, tuple(tuple(x.shape) for x in feats), feats[0].device,
               ops.stream_id())
        a = self._bufs.get(key)
        if a is None:
            # ping-pong activations of ONE pad shape (4 names); real evaluation sees many pad shapes
            # (keep-ratio 1333 x 800 padded to /32): keep the two most recent shapes' buffers, the
            # caching allocator recycles the rest (the cache grew without bound, ~1.1 GB
            # per shape at batch 16)
            while len(self._bufs) >= 8:
                self._bufs.popitem(last=False)
            a = self._bufs[key] = [torch.empty((x.shape[0], channels, x.shape[2], x.shape[3]),
                                               dtype=torch.bfloat16, device=x.device,
                                               memory_format=torch.channels_last) for x in feats]
        else:
            self._bufs.move_to_end(key)
        return a

    def __call__(self, feats):
        """feats: per-level (B, Cin, H, W) channels-last bf16 -> (cls[L], reg[L], iou[L])"""
        F = self.F
        feats = list(feats)
        halves = lambda ts: [[t[:, :F] for t in ts], [t[:, F:] for t in ts]]  # noqa: E731
        cur = self._acts('a', feats, 2 * F)
        ops.conv3x3_bf16_levels([feats], self.w0, self.b0, 2 * F, [cur], relu=True)
        if self.n_layers == 1:
            cls_feat, reg_feat = [t[:, :F] for t in cur], [t[:, F:].contiguous(memory_format=torch.channels_last) for t in cur]
        for i, (w, b) in enumerate(zip(self.w, self.b)):
            last = i == len(self.w) - 1
            if last:
                # the last tower layer writes two dense F-channel tensors: retina_reg / retina_iou
                # are library convolutions and want a dense input
                cls_feat, reg_feat = self._acts('c', feats, F), self._acts('r', feats, F)
                ops.conv3x3_bf16_levels(halves(cur), w, b, F, [cls_feat, reg_feat], relu=True)
            else:
                nxt = self._acts('b' if cur is self._acts('a', feats, 2 * F) else 'a', feats, 2 * F)
                ops.conv3x3_bf16_levels(halves(cur), w, b, F, halves(nxt), relu=True)
                cur = nxt
        cls = [torch.empty((x.shape[0], self.c_cls, x.shape[2], x.shape[3]), dtype=torch.bfloat16,
                           device=x.device, memory_format=torch.channels_last) for x in feats]
        ops.conv3x3_bf16_levels([cls_feat], self.w_cls, self.b_cls, self.c_cls, [cls], relu=False)
        ri = [torch.empty((x.shape[0], self.c_ri, x.shape[2], x.shape[3]), dtype=torch.bfloat16,
                          device=x.device, memory_format=torch.channels_last) for x in feats]
        ops.conv3x3_bf16_levels([reg_feat], self.w_ri, self.b_ri, self.c_ri, [ri], relu=False)
        cl = torch.channels_last
        reg = [t[:, :self.c_reg].contiguous(memory_format=cl) for t in ri]
        iou = [t[:, self.c_reg:self.c_reg + self.c_iou].contiguous(memory_format=cl) for t in ri]
        return cls, reg, iou
