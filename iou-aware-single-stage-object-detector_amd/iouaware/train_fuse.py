"""Training forward / backward of the ResNet bottlenecks and the FPN on the library's own
convolution paths (BASELINE config 5: the R-50 training iteration; reference resnet.py:215-255
for the block, fpn.py:100-136 for the neck).

In the reference's training configuration the backbone's BatchNorm layers run in eval mode
(`norm_eval=True`: frozen statistics, trainable gamma / beta), so conv + BN is an affine map
of the convolution output whose scale folds into the weight:

    y = relu( conv(x, w * s) + b  [+ identity] ),    s = gamma / sqrt(var + eps),  b = beta - mean * s

PyTorch eager runs that as MIOpen convolution (NCHW Winograd F(2x2) kernels fed through layout
transposes) + BatchNorm + add + ReLU forward, and their four backward kernels, every one a full
pass over the activation.  Here every convolution of a block is ONE autograd node over the
channels-last activation:

    1x1            `Conv1x1`: hipBLASLt GEMM with bias / identity / ReLU in the epilogue
                   (ops.linear_bias_act, the inference kernel, the (Cout, Cin) weight read
                   through the transpose flag); backward = ReLU mask + bias gradient in one
                   pass (csrc/trainops.hip), dx = g . w (the same GEMM entry point),
                   dw = g^T . x cut into slices of the pixel dimension (a (Cout x Cin) result
                   alone would occupy a handful of compute units)
    3x3, stride 1  winograd_train.wino_conv_levels: F(4x4,3x3) forward, input gradient and
                   Winograd-domain weight gradient
    3x3, stride 2  (three per network) torch's convolution on the folded weight

The fold `w * s`, `beta - mean * s` is one more autograd node on the parameters (`FoldBN`, one
kernel forward, one backward), which carries d(w*s), db back to w, gamma and beta; nothing of
BatchNorm's own backward is left.
Blocks whose parameters are all frozen (`frozen_stages`) and whose input carries no gradient take
the inference route of fuse.py.  Everything here needs the HIP library; there is no fallback
inside -- `usable()` decides before anything runs, and a module that does not qualify runs its
ordinary forward.
"""
import torch
import torch.nn.functional as F
from torch.nn.modules.batchnorm import _BatchNorm

from . import ops
from . import winograd_train as WT


# ------------------------------------------------------------------ 1x1 convolution node
def _split_for(P, k, n):
    """how many slices of the pixel dimension the weight-gradient GEMM is cut into: enough
    (k x n) result tiles to cover the 256 compute units a few times, slices >= 512 pixels"""
    tiles = max(1, (k // 128 or 1) * (n // 128 or 1))
    want = max(1, min(1024 // tiles, P // 512))
    s = 1
    for d in range(want, 0, -1):
        if P % d == 0:
            s = d
            break
    return s


def weight_grad_1x1(x2, g2):
    """x2 (P, k), g2 (P, n) -> x2^T g2 (k, n), the reduction over P cut into slices"""
    P, k = x2.shape
    n = g2.shape[1]
    s = _split_for(P, k, n)
    if s == 1:
        return x2.t().mm(g2)
    part = torch.bmm(x2.view(s, P // s, k).transpose(1, 2), g2.view(s, P // s, n))
    return part.sum(0)


def _rows(t):
    """channels-last (B, C, H, W) -> (B*H*W, C) view"""
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C)


def _cl(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) \
        else t.contiguous(memory_format=torch.channels_last)


# 'bmm': weight_grad_1x1, the reduction over the pixels cut into slices (batched GEMM + sum);
# 'lt': ops.gemm_tn, one library GEMM (its timed candidates do not split the reduction well:
# 38.7 vs 36.9 ms per R-50 iteration)
WGRAD = 'bmm'


class Conv1x1(torch.autograd.Function):
    """relu?( x . w^T + bias + identity ) on channels-last fp32 activations; w: (Cout, Cin)"""

    @staticmethod
    def forward(ctx, x, w_nk, bias, identity, relu):
        x = _cl(x)
        w = w_nk.contiguous()
        y = ops.linear_bias_act(x, w, None if bias is None else bias.contiguous(),
                                residual=None if identity is None else _cl(identity), relu=relu,
                                w_nk=True)
        ctx.relu = bool(relu)
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        need_x, need_w, need_b, need_i = ctx.needs_input_grad[:4]
        g, db = WT.relu_bwd_bias_grad(dy, y if ctx.relu else None, need_b)
        dx = dw = None
        if need_x:
            dx = ops.linear_bias_act(g, w, None)          # (rows, Cout) . (Cout, Cin)
        if need_w:
            g2, x2 = _rows(g), _rows(x)
            dw = ops.gemm_tn(g2, x2) if WGRAD == 'lt' else weight_grad_1x1(g2, x2)
        return dx, dw, db, (g if need_i else None), None


def conv1x1(x, w_nk, bias=None, identity=None, relu=False):
    return Conv1x1.apply(x, w_nk, bias, identity, relu)


class Conv1x1Fork(torch.autograd.Function):
    """conv1 of a bottleneck at the residual fork: (relu?(x . w^T + bias), x) -- x is handed on as
    the block's identity branch, so that BOTH gradients of the block input arrive in this node
    and the sum `dx = g . w + d_identity` rides in the input-gradient GEMM's epilogue (autograd
    would add the two contributions in a separate pass over the block's largest tensor)"""

    @staticmethod
    def forward(ctx, x, w_nk, bias, relu):
        x = _cl(x)
        w = w_nk.contiguous()
        y = ops.linear_bias_act(x, w, None if bias is None else bias.contiguous(), relu=relu,
                                w_nk=True)
        ctx.relu = bool(relu)
        ctx.save_for_backward(x, w, y if relu else None)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, d_idn):
        x, w, y = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        dx = dw = db = None
        if dy is None:                                   # only the identity branch was used
            return (d_idn if need_x else None), None, None, None
        g, db = WT.relu_bwd_bias_grad(dy, y if ctx.relu else None, need_b)
        if need_x:
            dx = ops.linear_bias_act(g, w, None, residual=None if d_idn is None else _cl(d_idn))
        if need_w:
            dw = ops.gemm_tn(_rows(g), _rows(x)) if WGRAD == 'lt' else weight_grad_1x1(_rows(g), _rows(x))
        return dx, dw, db, None


class Subsample(torch.autograd.Function):
    """x[:, :, ::s, ::s] of a channels-last activation as a channels-last tensor (the input of a
    strided 1x1 shortcut convolution); the gradient comes back channels-last as well (torch's
    slice backward builds an NCHW zero tensor that would then be converted)"""

    @staticmethod
    def forward(ctx, x, s):
        ctx.shape, ctx.s = tuple(x.shape), int(s)
        return x[:, :, ::s, ::s].contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, g):
        dx = torch.empty(ctx.shape, dtype=g.dtype, device=g.device,
                         memory_format=torch.channels_last).zero_()
        dx[:, :, ::ctx.s, ::ctx.s] = g
        return dx, None


def conv1x1_fork(x, w_nk, bias=None, relu=False):
    return Conv1x1Fork.apply(x, w_nk, bias, relu)


# ------------------------------------------------------------------ eval-mode BatchNorm fold
def _inv_std(bn):
    key = (bn.running_var.data_ptr(), bn.running_var._version)
    inv = getattr(bn, '_ia_inv', None)
    if inv is None or inv[0] != key:
        with torch.no_grad():
            inv = bn._ia_inv = (key, torch.rsqrt(bn.running_var.float() + bn.eps).contiguous())
    return inv[1]


def _dense_rows(w):
    """every output channel's Cin*kh*kw weights dense in memory (contiguous / channels-last)"""
    return w.is_contiguous() or (w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last))


def _same_layout(a, b):
    cl = torch.channels_last
    return (a.is_contiguous() and b.is_contiguous()) or \
        (a.dim() == 4 and a.is_contiguous(memory_format=cl) and b.is_contiguous(memory_format=cl))


class FoldBN(torch.autograd.Function):
    """(w, gamma, beta) -> (w * s, beta - mean * s), s = gamma / sqrt(var + eps): eval-mode
    BatchNorm folded into the convolution in front of it; one kernel forward, one backward
    (csrc/trainops.hip)"""

    @staticmethod
    def forward(ctx, w, gamma, beta, mean, inv):
        from . import _lib
        from .ops import _ptr, _stream
        cout = int(w.shape[0])
        wf = torch.empty_like(w)
        bf = torch.empty(cout, dtype=torch.float32, device=w.device)
        _lib.check(_lib.lib().ia_bn_fold_fwd(_ptr(w), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(inv),
                                             cout, w.numel() // cout, _ptr(wf), _ptr(bf), _stream()),
                   'ia_bn_fold_fwd')
        ctx.save_for_backward(w, gamma, mean, inv)
        return wf, bf

    @staticmethod
    def backward(ctx, dwf, dbf):
        from . import _lib
        from .ops import _ptr, _stream
        w, gamma, mean, inv = ctx.saved_tensors
        cout = int(w.shape[0])
        if dwf is None:                                   # only the shift was used
            dwf = torch.zeros_like(w)
        elif not _same_layout(dwf, w):
            dwf = torch.empty_like(w).copy_(dwf)
        dw = torch.empty_like(w) if ctx.needs_input_grad[0] else None
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma)
        _lib.check(_lib.lib().ia_bn_fold_bwd(_ptr(dwf), _ptr(None if dbf is None else dbf.contiguous()),
                                             _ptr(w), _ptr(gamma), _ptr(mean), _ptr(inv), cout,
                                             w.numel() // cout, _ptr(dw), _ptr(dgamma), _ptr(dbeta),
                                             _stream()), 'ia_bn_fold_bwd')
        return dw, dgamma, dbeta, None, None


def fold_bn(conv, bn):
    """conv weight and eval-mode BatchNorm -> (folded weight, shift), differentiable w.r.t. the
    weight, gamma and beta"""
    w = conv.weight
    inv = _inv_std(bn)
    if bn.weight is not None and bn.bias is not None and w.is_cuda and w.dtype == torch.float32 \
            and bn.weight.dtype == torch.float32 and _dense_rows(w):
        return FoldBN.apply(w, bn.weight, bn.bias, bn.running_mean, inv)
    s = inv if bn.weight is None else bn.weight * inv             # BatchNorm without affine part
    shift = -(bn.running_mean * s) if bn.bias is None \
        else torch.addcmul(bn.bias, bn.running_mean, s, value=-1.0)
    return w * s.view(-1, 1, 1, 1), shift


def _nk(w):
    """(Cout, Cin, 1, 1) -> (Cout, Cin)"""
    return w.view(w.shape[0], w.shape[1])


class _ParamMatrix(torch.autograd.Function):
    """a (Cout, Cin, 1, 1) PARAMETER as its (Cout, Cin) matrix; the gradient goes back with the
    parameter's own strides (channels-last parameters: DistributedDataParallel's bucket views
    follow the parameter's layout and would otherwise restride every gradient)"""

    @staticmethod
    def forward(ctx, w):
        ctx.size, ctx.stride = w.size(), w.stride()
        return w.view(w.shape[0], w.shape[1])

    @staticmethod
    def backward(ctx, g):
        return g.contiguous().as_strided(ctx.size, ctx.stride)


def _gemm_ok(conv, stride_ok=(1,)):
    return (tuple(conv.kernel_size) == (1, 1) and conv.stride[0] == conv.stride[1]
            and conv.stride[0] in stride_ok and tuple(conv.padding) == (0, 0)
            and conv.groups == 1 and conv.bias is None)


def _eval_bn(m):
    return isinstance(m, _BatchNorm) and not m.training and m.running_var is not None


def _act_ok(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4


def frozen(module, x):
    return (not x.requires_grad) and not any(p.requires_grad for p in module.parameters())


def bottleneck_usable(m, x):
    c2 = m.conv2
    return (torch.is_grad_enabled() and _act_ok(x) and not m.with_cp
            and _eval_bn(m.norm1) and _eval_bn(m.norm2) and _eval_bn(m.norm3)
            and _gemm_ok(m.conv1) and _gemm_ok(m.conv3)
            and tuple(c2.kernel_size) == (3, 3) and tuple(c2.padding) == (1, 1)
            and tuple(c2.dilation) == (1, 1) and c2.groups == 1 and c2.bias is None
            and c2.stride[0] == c2.stride[1] and c2.stride[0] in (1, 2)
            and c2.in_channels % 4 == 0 and c2.out_channels % 4 == 0
            and (m.downsample is None or (_gemm_ok(m.downsample[0], (1, 2))
                                          and _eval_bn(m.downsample[1]))))


def bottleneck_forward(m, x):
    """Bottleneck.forward (reference resnet.py:215-255) in training, eval-mode BatchNorm"""
    x = _cl(x)
    w1, b1 = fold_bn(m.conv1, m.norm1)
    w2, b2 = fold_bn(m.conv2, m.norm2)
    w3, b3 = fold_bn(m.conv3, m.norm3)
    if x.requires_grad:
        out, x = conv1x1_fork(x, _nk(w1), b1, True)     # x: the identity branch from here on
    else:
        out = conv1x1(x, _nk(w1), b1, None, True)
    if m.conv2.stride[0] == 1:
        out = WT.wino_conv_levels([out], w2, b2, relu=True)[0]
    else:
        out = F.relu(F.conv2d(out, w2, b2, m.conv2.stride, m.conv2.padding))
    if m.downsample is None:
        idn = x
    else:
        ds = m.downsample[0]
        wd, bd = fold_bn(ds, m.downsample[1])
        xs = x if ds.stride[0] == 1 else Subsample.apply(x, ds.stride[0])
        idn = conv1x1(xs, _nk(wd), bd, None, False)
    return conv1x1(out, _nk(w3), b3, idn, True)


# ------------------------------------------------------------------ FPN
def fpn_usable(m, inputs):
    if not (torch.is_grad_enabled() and len(inputs) == len(m.in_channels)
            and m.out_channels % 4 == 0):
        return False
    used = inputs[m.start_level:m.backbone_end_level]
    if not all(_act_ok(t) and t.shape[1] % 4 == 0 for t in used):
        return False
    for lc in m.lateral_convs:
        c = lc.conv
        if lc.with_norm or lc.with_activatation or tuple(c.kernel_size) != (1, 1) \
                or tuple(c.stride) != (1, 1) or tuple(c.padding) != (0, 0) or c.groups != 1:
            return False
    for fc in list(m.fpn_convs)[:len(m.lateral_convs)]:
        c = fc.conv
        if fc.with_norm or fc.with_activatation or tuple(c.kernel_size) != (3, 3) \
                or tuple(c.stride) != (1, 1) or tuple(c.padding) != (1, 1) or c.groups != 1:
            return False
    return True


def fpn_forward(m, inputs):
    """FPN.forward (reference fpn.py:100-136): laterals as GEMMs with the upsampled coarser level
    as the epilogue's identity operand, output convolutions on the Winograd path"""
    n = len(m.lateral_convs)
    lat = [None] * n
    for i in range(n - 1, -1, -1):
        c = m.lateral_convs[i].conv
        x = _cl(inputs[i + m.start_level])
        idn = None
        if i < n - 1:
            idn = F.interpolate(lat[i + 1], scale_factor=2, mode='nearest')
            if tuple(idn.shape[-2:]) != tuple(x.shape[-2:]):
                raise RuntimeError('FPN levels are not a factor of two apart')   # as the reference
        lat[i] = conv1x1(x, _ParamMatrix.apply(c.weight), c.bias, idn, False)
    outs = []
    for i in range(n):
        c = m.fpn_convs[i].conv
        outs.append(WT.wino_conv_levels([lat[i]], c.weight, c.bias, relu=False)[0])
    if m.num_outs > n:
        if not m.add_extra_convs:
            for _ in range(m.num_outs - n):
                outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        else:
            first = inputs[m.backbone_end_level - 1] if m.extra_convs_on_inputs else outs[-1]
            outs.append(m.fpn_convs[n](first))
            for i in range(n + 1, m.num_outs):
                src = F.relu(outs[-1]) if m.relu_before_extra_convs else outs[-1]
                outs.append(m.fpn_convs[i](src))
    return tuple(outs)
