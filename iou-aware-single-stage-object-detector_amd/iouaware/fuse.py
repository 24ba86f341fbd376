"""Inference-time fusion of the elementwise chains around the convolutions.

`fuse_inference(model)` switches ResNet/ResNeXt blocks, the stem and every
ConvModule to a forward that runs  conv (MIOpen)  ->  ONE HIP epilogue kernel
(csrc/elementwise.hip: per-channel affine [+ residual affine] [+ ReLU], in
place) instead of PyTorch eager's BatchNorm, add and ReLU kernels, each of
which is a full read+write pass over the activation:

    Bottleneck (reference resnet.py:215-255):  bn1,relu | bn2,relu | bn3,(bn_d),add,relu
                                               7-8 passes  ->  3 passes
    ConvModule (conv_module.py:149-163):       bias add, relu  ->  1 pass

Parameters and state_dict are untouched (checkpoints still load); the folded
per-channel scale/shift and weight copies are derived from them by `fuse_inference`
and derived AGAIN whenever the source tensors change (stamp check in every fused
forward: checkpoint loads, optimizer steps, `.to()` are all safe after fusing; and after every
training-mode / grad-mode forward, because torch's fused optimizers update parameters without
bumping the version counters the stamp reads).  Only eval-mode, no-grad,
GPU forwards take the fused route; anything else falls back to the module's
ordinary forward.  BatchNorm in eval mode is y = (x-mean)/sqrt(var+eps)*g + b;
the folded form x*scale+shift differs from it by rounding only.
"""
import types

import torch
import torch.nn.functional as F
from torch.nn.modules.batchnorm import _BatchNorm

from . import ops, train_fuse
from .backbones import BasicBlock, Bottleneck, ResNet
from .layers import ConvModule

# channels per 16-byte vector of the fused elementwise kernels
_VEC = {torch.float32: 4, torch.bfloat16: 8}


def _fold_bn(bn):
    if not isinstance(bn, _BatchNorm):
        raise TypeError('only BatchNorm can be folded, got %s' % type(bn).__name__)
    with torch.no_grad():
        var, mean = bn.running_var.float(), bn.running_mean.float()
        g = bn.weight.float() if bn.weight is not None else torch.ones_like(var)
        b = bn.bias.float() if bn.bias is not None else torch.zeros_like(var)
        scale = g / torch.sqrt(var + bn.eps)
        shift = b - mean * scale
    return scale.contiguous(), shift.contiguous()


def _stamp_slots(module):
    """where the tensors of `_stamp` live: (parameter / buffer dict, key) pairs -- a dict lookup per
    forward instead of a walk over the submodules through nn.Module.__getattr__ (32 us per fused module
    and forward: a third of a batch-1 step's host time, which is what bounds that step) -- and where the
    modules that own them hang: (parent's `_modules` dict, name, module) triples, so that a submodule
    REPLACED after fuse_inference (`blk.conv2 = other_conv`) is noticed (the slots alone kept
    pointing into the old module's dicts and the stale folded weights were used silently)."""
    slots, links = [], []
    # a ResNet's own folded copies derive from its stem only (the blocks below are fused modules
    # with their own stamps): do not re-fold the frozen stem because a trainable stage took an
    # optimizer step
    tops = ('conv1', module.norm1_name) if isinstance(module, ResNet) else (None,)
    for top_name in tops:
        top = module if top_name is None else module._modules[top_name]
        if top_name is not None:
            links.append((module._modules, top_name, top))
        stack = [top]
        while stack:
            m = stack.pop()
            for name, child in m._modules.items():
                if child is not None:
                    links.append((m._modules, name, child))
                    stack.append(child)
            if isinstance(m, torch.nn.Conv2d):
                slots.append((m._parameters, 'weight'))
                if m.bias is not None:
                    slots.append((m._parameters, 'bias'))
            elif isinstance(m, _BatchNorm) and m.running_var is not None:
                slots.append((m._buffers, 'running_var'))
                if m.weight is not None:
                    slots.append((m._parameters, 'weight'))
    return slots, links


def _stamp(module):
    """identity + version of what the folded copies were derived from: every convolution weight
    and one running statistic per BatchNorm under `module`.  An optimizer step, a checkpoint
    load (in-place copies bump `_version`) or `.to(device / dtype)` (new storage, also a new buffer
    object: the slots are looked up in the modules' own dicts every time) changes it.  A submodule
    that was replaced (or removed) since the slots were collected rebuilds them: the stamp then differs
    from the stored one because the new module's tensors are other tensors."""
    cached = module.__dict__.get('_ia_stamp_slots')
    if cached is not None:
        for d, name, child in cached[1]:
            if d.get(name) is not child:
                cached = None
                break
    if cached is None:
        cached = module.__dict__['_ia_stamp_slots'] = _stamp_slots(module)
    out = []
    for d, k in cached[0]:
        t = d[k]
        out.append((t.data_ptr(), t._version) if t is not None else (0, 0))
    return tuple(out)


def _mark_dirty(module):
    """this module and every fused module below it: their derived copies must be refreshed before
    the next inference forward (a training forward ran; see _fast)"""
    for m in module.modules():
        if hasattr(m, '_ia_opts'):
            m._ia_dirty = True


def _fast(module, x):
    # no autograd graph to build: grad mode off, or a frozen module (frozen_stages) fed an input
    # that carries no gradient -- then the raw kernels' outputs (requires_grad False) are what
    # eager would have produced as well
    ok = (not module.training) and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) \
        and (not torch.is_grad_enabled() or train_fuse.frozen(module, x))
    if not ok:
        if module.training or torch.is_grad_enabled():
            # a forward that may be followed by an optimizer step.  The stamp below sees in-place
            # updates through the tensors' version counters, but torch's FUSED optimizers
            # (torch.optim.SGD(fused=True), ...) write the parameters without bumping them:
            # derive the copies again at the next inference forward, whatever the stamp says
            module._ia_dirty = True
        return False
    if getattr(module, '_ia_dirty', False) or module._ia_stamp != _stamp(module):
        _fold(module)              # parameters changed since the fold: derive the copies again
    return True


def _conv_nobias(conv, x):
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


def _train(module):
    return len(module._ia_opts) > 2 and module._ia_opts[2]


def _bottleneck_forward(self, x):
    return _bottleneck_run(self, x)[0]


def _bottleneck_run(self, x, h1=None, nxt=None):
    """-> (y, h).  h1: this block's conv1 + bn1 + ReLU output, already computed by the block in
    front (chained stage-1 boundary, ops.conv1x1_chain); nxt: the next block's folded copies -- the
    tail then produces y AND the next block's h1 in one kernel (h is None when it did not).  Both
    only from _layer_forward, which has checked the shapes."""
    if not _fast(self, x):
        if _train(self) and train_fuse.bottleneck_usable(self, x):
            return train_fuse.bottleneck_forward(self, x), None      # training, eval-mode BatchNorm
        return type(self).forward(self, x), None
    f = self._ia_fused
    lt = 'w1' in f and x.dtype in (torch.float32, torch.bfloat16) \
        and x.is_contiguous(memory_format=torch.channels_last)
    if h1 is not None:
        w, out, pre = f, h1, None
    elif lt:
        # 1x1 convolutions = library GEMMs on the channels-last activation, folded BN / residual /
        # ReLU in the GEMM epilogue (csrc/gemm.hip); bf16 networks use bf16 copies of the folded
        # weights (biases stay fp32, accumulation is fp32)
        w = _weights_for(f, x.dtype)
        out, pre = ops.linear_bias_act(x, w['w1'], f['b1'], relu=True), None
    else:
        out, pre = self.conv1(x), (f['s1'], f['b1'], True)
    gw = f.get('gconv2')
    wino = f.get('wino2')
    if gw is not None and out.dtype == torch.float32 \
            and out.is_contiguous(memory_format=torch.channels_last):
        # ResNeXt: grouped 3x3 conv2 + folded BN + ReLU on the MFMA kernel of csrc/gconv.hip
        if pre is not None:
            out = ops.channel_affine_act_(out, f['s1'], f['b1'], relu=True)
        out = ops.grouped_conv3x3(out, gw, f['b2'], self.conv2.groups, self.conv2.stride[0],
                                  relu=True)
    elif wino is not None and wino.usable(out):
        # (conv1's BN + ReLU on load,) conv2 + its folded BN + ReLU in the Winograd path
        out = wino(out, pre=pre)
    elif pre is None and 'c3' in f and out.dtype == torch.bfloat16 \
            and out.is_contiguous(memory_format=torch.channels_last):
        # bf16 (BASELINE config 3): conv2 + folded BN + ReLU on this library's MFMA implicit-GEMM
        # kernel (csrc/conv3x3_bf16.hip; weights packed once per fold, BN scale folded in)
        wp = f.get('_c3_packed')
        if wp is None:
            wp = f['_c3_packed'] = ops.conv3x3_bf16_pack(f['c3'])
        out = ops.conv3x3_bf16(out, wp, f['b2'], self.conv2.out_channels, relu=True)
    elif lt and 'im2col2' in f and out.is_contiguous(memory_format=torch.channels_last):
        # stride-2 conv2 (first block of stages 2-4): im2col + library GEMM, folded BN + ReLU in
        # the epilogue -- a fixed reduction order, where the library convolution's split-K kernels
        # add with atomics (csrc/im2col.hip)
        out = ops.conv3x3_im2col(out, w['im2col2'], f['b2'], stride=self.conv2.stride[0], relu=True)
    else:
        if pre is not None:
            out = ops.channel_affine_act_(out, f['s1'], f['b1'], relu=True)
        out = ops.channel_affine_act_(self.conv2(out), f['s2'], f['b2'], relu=True)
    if lt and out.is_contiguous(memory_format=torch.channels_last):
        if self.downsample is None:
            if nxt is not None:
                return ops.conv1x1_chain(out, w['w3'], f['b3'], x, nxt['w1'], nxt['b1'])
            return ops.linear_bias_act(out, w['w3'], f['b3'], residual=x, relu=True), None
        if 'wd' in f:                     # stride-1 projection: a GEMM as well
            idn = ops.linear_bias_act(x, w['wd'], f['b3d'])
            if nxt is not None:
                return ops.conv1x1_chain(out, w['w3'], None, idn, nxt['w1'], nxt['b1'])
            return ops.linear_bias_act(out, w['w3'], None, residual=idn, relu=True), None
        ds = self.downsample[0]
        if 'wd_s' in f:                   # strided projection: one strided-batched GEMM, input read in place
            idn = ops.conv1x1_strided(x, w['wd_s'], None, None, stride=ds.stride[0])
        else:
            idn = F.conv2d(x, w['wd_conv'], None, ds.stride, ds.padding)
        return ops.linear_bias_act(out, w['w3'], f['b3d'], residual=idn, relu=True), None
    out = self.conv3(out)
    if self.downsample is None:
        return ops.channel_affine_act_(out, f['s3'], f['b3'], residual=x, relu=True), None
    idn = self.downsample[0](x)
    return ops.channel_affine_act_(out, f['s3'], f['b3'], residual=idn, res_scale=f['sd'],
                                   res_shift=f['bd'], relu=True), None


def _chain_ok(blk, nxt, x):
    """can `blk`'s conv3 + add + ReLU and `nxt`'s conv1 + ReLU run as one kernel on input x?"""
    f, g = getattr(blk, '_ia_fused', None), getattr(nxt, '_ia_fused', None)
    if f is None or g is None or 'w3' not in f or 'w1' not in g or nxt.downsample is not None:
        return False
    if blk.downsample is not None and 'wd' not in f:
        return False
    k, n = f['w3'].shape
    return tuple(g['w1'].shape) == (n, g['w1'].shape[1]) and ops.chain_usable(x, k, n, g['w1'].shape[1]) \
        and x.is_contiguous(memory_format=torch.channels_last)


def _has_hooks(m):
    """forward / forward-pre hooks on the module (or registered globally): a chained block is run by
    _bottleneck_run directly, not through nn.Module.__call__, so its hooks would not fire"""
    import torch.nn.modules.module as tm
    return bool(m._forward_hooks or m._forward_pre_hooks or tm._global_forward_hooks
                or tm._global_forward_pre_hooks)


def _layer_forward(self, x):
    """a residual stage (nn.Sequential of bottlenecks, resnet.py:106-123): the blocks one after the
    other, and where two consecutive blocks allow it (stage 1 at the benchmark sizes) the boundary
    between them in ONE kernel -- the tail of block i also produces conv1 of block i + 1"""
    blocks = list(self)
    if not all(isinstance(b, Bottleneck) and hasattr(b, '_ia_opts') for b in blocks) or not blocks:
        return torch.nn.Sequential.forward(self, x)
    h1 = None
    for i, blk in enumerate(blocks):
        nb = blocks[i + 1] if i + 1 < len(blocks) else None
        # (_fast(nb, .) looks at dtype / device / memory format only -- block i + 1 receives a tensor
        # with the same three as x; the second _chain_ok: a re-fold inside _fast replaces the dicts.
        # Blocks with hooks are never bypassed: feature taps and profilers keep firing)
        chain = nb is not None and x.dtype == torch.float32 and not _has_hooks(blk) and not _has_hooks(nb) \
            and _chain_ok(blk, nb, x) and _fast(blk, x) and _fast(nb, x) and _chain_ok(blk, nb, x)
        if chain or h1 is not None:
            x, h1 = _bottleneck_run(blk, x, h1=h1, nxt=nb._ia_fused if chain else None)
        else:
            x = blk(x)
    return x


def _weights_for(f, dtype):
    """the folded GEMM weights of a fused module in the activation dtype (fp32 originals; other
    dtypes are cast once and cached)"""
    if dtype == torch.float32:
        return f
    cache = f.setdefault('_cast', {})
    w = cache.get(dtype)
    if w is None:
        w = cache[dtype] = {k: v.to(dtype) for k, v in f.items()
                            if k in ('w1', 'w3', 'wd', 'wd_conv', 'wd_s', 'w_kn', 'im2col2', 'im2col')}
    return w


def _basic_forward(self, x):
    if not _fast(self, x):
        return type(self).forward(self, x)
    f = self._ia_fused
    out = ops.channel_affine_act_(self.conv1(x), f['s1'], f['b1'], relu=True)
    out = self.conv2(out)
    if self.downsample is None:
        return ops.channel_affine_act_(out, f['s2'], f['b2'], residual=x, relu=True)
    idn = self.downsample[0](x)
    return ops.channel_affine_act_(out, f['s2'], f['b2'], residual=idn, res_scale=f['sd'],
                                   res_shift=f['bd'], relu=True)


def _resnet_forward(self, x):
    if not _fast(self, x):
        if _train(self) and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 \
                and not x.requires_grad and not self.norm1.training \
                and not any(p.requires_grad for m in (self.conv1, self.norm1)
                            for p in m.parameters()):
            # training with a frozen stem (frozen_stages >= 0): the stem on the inference kernels,
            # the stages decide for themselves (frozen -> inference route, else train_fuse)
            if self._ia_stamp != _stamp(self):
                _fold(self)
            with torch.no_grad():
                x = self._stem(x.contiguous(memory_format=torch.channels_last))
            return self._stages(x)
        return type(self).forward(self, x)
    return self._stages(self._stem(x))


def _resnet_stem(self, x):
    f = self._ia_fused
    if 'stem_w' in f and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 \
            and x.is_contiguous(memory_format=torch.channels_last):
        x = ops.stem_conv(x, f['stem_w'])         # own fp32 MFMA kernel (csrc/stem.hip), raw convolution
    elif 'stem_w' in f and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] == 3 \
            and x.is_contiguous(memory_format=torch.channels_last):
        w16 = f.get('_stem_w16')
        if w16 is None:
            w16 = f['_stem_w16'] = ops.stem_weight_bf16(self.conv1.weight)
        x = ops.stem_conv_bf16(x, w16)            # bf16 MFMA variant
    else:
        x = self.conv1(x)
    mp = self.maxpool
    if f.get('pool') and x.dtype in _VEC and x.shape[1] % _VEC[x.dtype] == 0 \
            and x.is_contiguous(memory_format=torch.channels_last):
        x = ops.affine_relu_maxpool(x, f['s'], f['b'])          # BN + ReLU + 3x3/2 max-pool, one pass
    else:
        x = mp(ops.channel_affine_act_(x, f['s'], f['b'], relu=True))
    return x


def _resnet_stages(self, x):
    outs = []
    for i, name in enumerate(self.res_layers):
        x = getattr(self, name)(x)
        if i in self.out_indices:
            outs.append(x)
    return tuple(outs)


def _convmodule_forward(self, x, activate=True, norm=True):
    if not (_fast(self, x) and self.activate_last):
        return type(self).forward(self, x, activate, norm)
    f = self._ia_fused
    relu = bool(activate and self.with_activatation)
    wino = f.get('wino')
    if wino is not None and wino.relu == relu and wino.usable(x):
        return wino(x)
    if 'c3' in f and norm and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last):
        # bf16 FPN output convolutions: the MFMA implicit-GEMM kernel, bias (+ReLU) in its epilogue
        wp = f.get('_c3_packed')
        if wp is None:
            wp = f['_c3_packed'] = ops.conv3x3_bf16_pack(f['c3'])
        return ops.conv3x3_bf16(x, wp, f.get('bias'), self.conv.out_channels, relu=relu)
    if 'w_kn' in f and norm and x.dtype in (torch.float32, torch.bfloat16) \
            and x.is_contiguous(memory_format=torch.channels_last):
        return ops.linear_bias_act(x, _weights_for(f, x.dtype)['w_kn'], f['b_kn'], relu=relu)
    if 'im2col' in f and x.dtype in (torch.float32, torch.bfloat16) \
            and x.is_contiguous(memory_format=torch.channels_last):
        return ops.conv3x3_im2col(x, _weights_for(f, x.dtype)['im2col'], f.get('bias'),
                                  stride=self.conv.stride[0], relu=relu)
    if self.with_norm and norm:
        y = self.conv(x)                      # conv before a norm has no bias
        return ops.channel_affine_act_(y, f['s'], f['b'], relu=relu)
    if self.conv.bias is None:
        y = self.conv(x)
        return ops.channel_affine_act_(y, None, None, relu=True) if relu else y
    return ops.channel_affine_act_(_conv_nobias(self.conv, x), None, f['bias'], relu=relu)


def _fpn_forward(self, inputs):
    """FPN.forward with the top-down `lat[i-1] + interpolate(lat[i])` as one in-place kernel"""
    # preconditions first: the fall-back recomputes everything, so nothing may run before it.
    # The lateral convolutions keep the layout / dtype of their inputs and have
    # out_channels outputs, which is all the fused top-down kernel needs to know.
    ok = (not self.training) and (not torch.is_grad_enabled()) \
        and len(inputs) == len(self.in_channels) and all(
            t.is_cuda and t.dtype in _VEC and t.dtype == inputs[0].dtype
            and self.out_channels % _VEC[t.dtype] == 0
            and t.is_contiguous(memory_format=torch.channels_last)
            for t in inputs[self.start_level:self.backbone_end_level])
    if not ok:
        if self.training or torch.is_grad_enabled():
            _mark_dirty(self)          # the ConvModules below are bypassed by the training route
        if _train(self) and train_fuse.fpn_usable(self, inputs):
            return train_fuse.fpn_forward(self, inputs)
        return type(self).forward(self, inputs)
    lat = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
    n = len(lat)
    if not all(t.is_contiguous(memory_format=torch.channels_last) for t in lat):
        lat = [t.contiguous(memory_format=torch.channels_last) for t in lat]
    for i in range(n - 1, 0, -1):
        if lat[i - 1].shape[2] == 2 * lat[i].shape[2] and lat[i - 1].shape[3] == 2 * lat[i].shape[3]:
            ops.upsample2x_add_(lat[i - 1], lat[i])
        else:
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], scale_factor=2, mode='nearest')
    outs = [self.fpn_convs[i](lat[i]) for i in range(n)]
    if self.num_outs > n:
        if not self.add_extra_convs:
            for _ in range(self.num_outs - n):
                outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        else:
            first = inputs[self.backbone_end_level - 1] if self.extra_convs_on_inputs else outs[-1]
            outs.append(self.fpn_convs[n](first))
            for i in range(n + 1, self.num_outs):
                src = F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]
                outs.append(self.fpn_convs[i](src))
    return tuple(outs)


def _head_forward(self, feats):
    w = self._ia_wino
    if (not self.training) and w.usable(feats):
        if getattr(self, '_ia_dirty', False) or self._ia_stamp != _stamp(self):
            _fold(self)
            w = self._ia_wino
        return w(list(feats))
    if (not self.training) and _bf16_head_ok(feats):
        # bf16 (BASELINE config 3): towers + class output on the MFMA implicit-GEMM kernel
        if getattr(self, '_ia_dirty', False) or self._ia_stamp != _stamp(self):
            _fold(self)
        c3 = getattr(self, '_ia_c3', None)
        if c3 is None:
            from .conv3x3_bf16 import Bf16ConvHead
            try:
                c3 = Bf16ConvHead(self)
            except (NotImplementedError, ValueError):
                c3 = False
            self._ia_c3 = c3
        if c3 and c3.usable(feats):
            return c3(feats)
    if self.training or torch.is_grad_enabled():
        _mark_dirty(self)              # see _fast: fused optimizers do not bump version counters
    return type(self).forward(self, feats)


def _bf16_head_ok(feats):
    return (not torch.is_grad_enabled()) and all(
        x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
        and x.is_contiguous(memory_format=torch.channels_last) for x in feats)


def _wino_ok(conv):
    return (tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1)
            and tuple(conv.padding) == (1, 1) and tuple(conv.dilation) == (1, 1)
            and conv.groups == 1 and conv.in_channels % 4 == 0 and conv.out_channels % 4 == 0)


def _im2col_ok(conv):
    """3x3 / pad 1 with a stride > 1: the im2col + GEMM route (csrc/im2col.hip)"""
    return (tuple(conv.kernel_size) == (3, 3) and conv.stride[0] == conv.stride[1]
            and 2 <= conv.stride[0] <= 4 and tuple(conv.padding) == (1, 1)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.in_channels % 8 == 0
            and getattr(conv, 'padding_mode', 'zeros') == 'zeros')


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _gemm_ok(conv):
    return (tuple(conv.kernel_size) == (1, 1) and tuple(conv.stride) == (1, 1)
            and tuple(conv.padding) == (0, 0) and conv.groups == 1 and conv.bias is None)


def _fold(m):
    """(re)derive the folded / transformed weight copies of one fused module from its current
    parameters.  Returns False for modules this file does not fuse."""
    winograd, fpn_conv = m._ia_opts[:2]
    if type(m).__name__ == 'FPN':
        return True                               # nothing folded: only the forward is replaced
    if type(m).__name__ == 'IoUawareRetinaHead':
        from .winograd import WinogradHead
        m._ia_wino = WinogradHead(m)
        m._ia_c3 = None                           # bf16 weights are packed on the first bf16 call
    elif isinstance(m, Bottleneck):
        f = {}
        f['s1'], f['b1'] = _fold_bn(m.norm1)
        f['s2'], f['b2'] = _fold_bn(m.norm2)
        f['s3'], f['b3'] = _fold_bn(m.norm3)
        if m.downsample is not None:
            f['sd'], f['bd'] = _fold_bn(m.downsample[1])
        if winograd and _gemm_ok(m.conv1) and _gemm_ok(m.conv3):
            with torch.no_grad():
                def kn(conv, scale):       # (Cout, Cin, 1, 1) * scale[Cout] -> (Cin, Cout)
                    w = conv.weight.float().view(conv.out_channels, conv.in_channels)
                    return (w * scale.view(-1, 1)).t().contiguous()
                f['w1'], f['w3'] = kn(m.conv1, f['s1']), kn(m.conv3, f['s3'])
                if m.downsample is not None:
                    ds = m.downsample[0]
                    f['b3d'] = (f['b3'] + f['bd']).contiguous()
                    if _gemm_ok(ds):
                        f['wd'] = kn(ds, f['sd'])
                    elif tuple(ds.kernel_size) == (1, 1) and tuple(ds.padding) == (0, 0) \
                            and ds.groups == 1 and ds.bias is None and ds.stride[0] == ds.stride[1]:
                        f['wd_s'] = kn(ds, f['sd'])
                    else:
                        f['wd_conv'] = (ds.weight.float() * f['sd'].view(-1, 1, 1, 1)).contiguous(
                            memory_format=torch.channels_last)
        c2 = m.conv2
        if winograd and c2.groups > 1 and tuple(c2.kernel_size) == (3, 3) \
                and tuple(c2.padding) == (1, 1) and tuple(c2.dilation) == (1, 1) \
                and c2.stride[0] == c2.stride[1] and c2.stride[0] in (1, 2) and c2.bias is None \
                and c2.in_channels == c2.out_channels \
                and c2.in_channels // c2.groups in (4, 8, 16, 32) \
                and c2.in_channels % (32 if c2.in_channels // c2.groups == 32 else 16) == 0:
            f['gconv2'] = ops.pack_grouped_weight(c2.weight, f['s2'])     # BN scale folded in
        if winograd and _im2col_ok(c2) and c2.bias is None and 'w1' in f:
            f['im2col2'] = ops.conv3x3_weight_kn(c2.weight, f['s2'])         # BN scale folded in
        if winograd and _wino_ok(m.conv2) and m.conv2.bias is None:
            from .winograd import WinogradConv3x3
            with torch.no_grad():           # BN scale folded into the weights, shift = bias
                w2 = m.conv2.weight.float() * f['s2'].view(-1, 1, 1, 1)
            f['wino2'] = WinogradConv3x3(w2, f['b2'], relu=True)
            if m.conv2.in_channels % 32 == 0 and m.conv2.out_channels % 2 == 0 \
                    and getattr(m.conv2, 'padding_mode', 'zeros') == 'zeros':
                f['c3'] = w2                # bf16 networks: packed for k_conv3x3_bf16 at the first bf16 call
        m._ia_fused = f
    elif isinstance(m, BasicBlock):
        f = {}
        f['s1'], f['b1'] = _fold_bn(m.norm1)
        f['s2'], f['b2'] = _fold_bn(m.norm2)
        if m.downsample is not None:
            f['sd'], f['bd'] = _fold_bn(m.downsample[1])
        m._ia_fused = f
    elif isinstance(m, ResNet):
        f = {}
        f['s'], f['b'] = _fold_bn(m.norm1)
        mp = m.maxpool
        f['pool'] = (_pair(mp.kernel_size), _pair(mp.stride), _pair(mp.padding),
                     _pair(mp.dilation), mp.ceil_mode) == ((3, 3), (2, 2), (1, 1), (1, 1), False)
        c1 = m.conv1
        if winograd and isinstance(c1, torch.nn.Conv2d) and tuple(c1.weight.shape) == (64, 3, 7, 7) \
                and tuple(c1.stride) == (2, 2) and tuple(c1.padding) == (3, 3) and tuple(c1.dilation) == (1, 1) \
                and c1.groups == 1 and c1.bias is None and getattr(c1, 'padding_mode', 'zeros') == 'zeros':
            f['stem_w'] = ops.stem_weight(c1.weight)
        m._ia_fused = f
    elif isinstance(m, ConvModule):
        f = {}
        c = m.conv
        if m.with_norm:
            if not isinstance(m.norm, _BatchNorm):
                return False                   # GroupNorm etc.: leave eager
            f['s'], f['b'] = _fold_bn(m.norm)
            if c.bias is not None:             # conv bias in front of a norm: (y + bias) * s + b
                with torch.no_grad():
                    f['b'] = (f['b'] + c.bias.detach().float() * f['s']).contiguous()
        elif c.bias is not None:
            f['bias'] = c.bias.detach().float().contiguous()
        if winograd and tuple(c.kernel_size) == (1, 1) and tuple(c.stride) == (1, 1) \
                and tuple(c.padding) == (0, 0) and c.groups == 1:
            with torch.no_grad():
                w = c.weight.float().view(c.out_channels, c.in_channels)
                if m.with_norm:
                    f['w_kn'], f['b_kn'] = (w * f['s'].view(-1, 1)).t().contiguous(), f['b']
                else:
                    f['w_kn'] = w.t().contiguous()
                    f['b_kn'] = None if c.bias is None else c.bias.detach().float().contiguous()
        if winograd and not m.with_norm and _im2col_ok(c):
            f['im2col'] = ops.conv3x3_weight_kn(c.weight)                    # P6 / P7
        if winograd and fpn_conv and not m.with_norm and _wino_ok(m.conv):
            from .winograd import WinogradConv3x3
            f['wino'] = WinogradConv3x3(m.conv.weight, m.conv.bias, relu=m.with_activatation)
            if c.in_channels % 32 == 0 and c.out_channels % 2 == 0 \
                    and getattr(c, 'padding_mode', 'zeros') == 'zeros':
                f['c3'] = c.weight.detach()
        m._ia_fused = f
    else:
        return False
    m.__dict__.pop('_ia_stamp_slots', None)         # (re)fold: the slots are rebuilt from the module as it is now
    m._ia_stamp = _stamp(m)
    m._ia_dirty = False
    return True


def _forward_for(m, winograd):
    if type(m).__name__ == 'FPN':
        return _fpn_forward if winograd else None
    if type(m).__name__ == 'IoUawareRetinaHead':
        return _head_forward if winograd else None
    for cls, fn in ((Bottleneck, _bottleneck_forward), (BasicBlock, _basic_forward),
                    (ResNet, _resnet_forward), (ConvModule, _convmodule_forward)):
        if isinstance(m, cls):
            return fn
    return None


def fuse_inference(model, winograd=False, train=False):
    """Patch `model` in place (see module docstring).  Returns the number of fused modules.

    train=True (with winograd=True) additionally gives the bottlenecks and the FPN a TRAINING
    route (train_fuse.py): with grad mode on and eval-mode BatchNorm (`norm_eval=True`), every
    convolution of a block is one autograd node on the GEMM / Winograd kernels; frozen stages
    take the inference route.

    winograd=True additionally routes the head's 3x3 convolutions through the Winograd
    F(4x4,3x3) path (iouaware/winograd.py) whenever its inputs are channels-last fp32 CUDA
    tensors: all pyramid levels in one batched GEMM per layer.

    The folded BatchNorm scale / shift, the GEMM-layout weights and the Winograd-transformed
    weights are COPIES of the parameters; every fused forward compares a stamp of the source
    tensors (storage pointer + version counter) and derives the copies again when a checkpoint
    load, an optimizer step or `.to()` changed them."""
    n = 0
    # FPN output convolutions (per-level weights) take the single-level Winograd path; the head's
    # ConvModules are bypassed by the head-level runner
    fpn_convs = set()
    if winograd:
        for m in model.modules():
            if type(m).__name__ == 'FPN':
                fpn_convs.update(id(c) for c in m.fpn_convs)
    for m in model.modules():
        fwd = _forward_for(m, winograd)
        if fwd is None:
            continue
        m._ia_opts = (bool(winograd), id(m) in fpn_convs, bool(train and winograd))
        if not _fold(m):
            del m._ia_opts
            continue
        if type(m).__name__ == 'FPN':
            m._ia_stamp = ()
        m.forward = types.MethodType(fwd, m)
        if isinstance(m, ResNet):
            m._stem = types.MethodType(_resnet_stem, m)
            m._stages = types.MethodType(_resnet_stages, m)
            if winograd:
                for name in m.res_layers:
                    layer = getattr(m, name)
                    layer._ia_layer = True
                    layer.forward = types.MethodType(_layer_forward, layer)
        n += 1
    return n


def refresh_fused(model):
    """derive the folded copies of every fused module again (after loading weights by means
    that bypass the stamp, e.g. `param.data = ...`).  Returns the number of modules touched."""
    n = 0
    for m in model.modules():
        if hasattr(m, '_ia_opts') and _fold(m):
            n += 1
    return n


def unfuse_inference(model):
    for m in model.modules():
        if hasattr(m, '_ia_layer'):
            del m._ia_layer
            m.__dict__.pop('forward', None)
        for attr in ('_ia_fused', '_ia_wino', '_ia_c3', '_ia_opts', '_ia_stamp', '_ia_dirty', '_ia_stamp_slots'):
            if hasattr(m, attr):
                delattr(m, attr)
                for name in ('forward', '_stem', '_stages'):
                    if name in m.__dict__:
                        del m.__dict__[name]
