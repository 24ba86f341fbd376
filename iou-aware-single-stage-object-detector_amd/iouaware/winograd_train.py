"""Training through the head's 3x3 convolutions on the Winograd F(4x4,3x3) path
(BASELINE config 5: the R-50 training step; reference iou_aware_retina_head.py:171-219 for the
convolutions, anchor_head.py / losses for what follows).

The IoU-aware head is 58 % of the network's multiply-adds, and in training every one of its
convolutions runs three times (forward, gradient w.r.t. the input, gradient w.r.t. the weights).
MIOpen serves the first two with its F(2x2,3x3) assembly kernels, one call per level and tower
(110 of the 139 Winograd calls of an iteration).  Here one shared-weight convolution over ALL
pyramid levels is one autograd node:

    forward     y_l = conv(x_l, w) + b  [ReLU]      in-transform -> 36 GEMMs -> out-transform
    grad input  dx_l = conv(dy_l, w^T flipped)      the same three steps with the transposed,
                                                    spatially flipped weight: full correlation
                                                    = the adjoint of a stride-1 / pad-1 conv
    grad weight dU[k] = V[k]^T dM[k], dW = G^T dU G  in the Winograd domain: V = B^T x B is kept
                                                    from the forward pass, dM = A dy A^T is one
                                                    more HIP transform (k_wino_dy), the 36
                                                    (Cin x tiles) x (tiles x Cout) products are one
                                                    batched library GEMM -- instead of one MIOpen
                                                    wrw call per level (17 of the 59 ms of an
                                                    iteration before)

-- the HIP transforms and the hipBLASLt batched GEMM of the inference path (winograd.py), 36
multiplications per 4x4 output tile instead of MIOpen's 64, and all levels in one GEMM.
Activations are channels-last fp32; the transformed weights are recomputed from the parameters
in every call (they change every iteration).
"""
import collections
import ctypes

import torch

from . import winograd as W


def _plan(xs):
    key = (xs[0].shape[0], tuple(tuple(x.shape[-2:]) for x in xs), xs[0].device,
           W.stream_id())
    return W._plan_for(_PLANS, key, lambda: W._Plan([tuple(x.shape[-2:]) for x in xs], xs[0].shape[0],
                                                    xs[0].device))


_PLANS = collections.OrderedDict()       # least recently used first, W._PLAN_ENTRIES kept (multi-scale training)


def _cl(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) \
        else t.contiguous(memory_format=torch.channels_last)


def _conv_levels(plan, xs, u, bias, relu, cout, tag, keep_v=False):
    cin = xs[0].shape[1]
    vbuf = torch.empty((36, plan.T, cin), dtype=torch.float32, device=xs[0].device) if keep_v \
        else plan.buf('tv' + tag, (36, plan.T, cin))
    v = W.input_transform(plan, xs, 1, vbuf)
    m = W.batched_gemm(v, u, plan.buf('tm' + tag, (36, plan.T, cout)))
    ys = [torch.empty((x.shape[0], cout) + tuple(x.shape[-2:]), dtype=torch.float32,
                      device=x.device, memory_format=torch.channels_last) for x in xs]
    W.output_transform(plan, m, cout, 1, bias, relu, [(0, cout, ys, 0)])
    return (ys, v) if keep_v else ys


def grad_output_transform(plan, dys, out):
    """dM = A dY A^T of every tile -> (36, T, Cout)"""
    import ctypes as C
    from . import _lib
    from .ops import _ptr, _stream
    ptrs = (C.c_void_p * len(dys))(*[d.data_ptr() for d in dys])
    ch = int(dys[0].shape[1])
    W._timed('dy', plan.T * ch * 4 * (16 + 36), lambda: _lib.check(
        _lib.lib().ia_wino_grad_output_transform(C.byref(plan.geom), ptrs, ch, _ptr(out),
                                                 _stream()), 'ia_wino_grad_output_transform'))
    return out


_G_HOST = (ctypes.c_double * 18)(*[float(v) for v in W._G.reshape(-1)])


def transform_weight(w, adjoint=False):
    """(Cout, Cin, 3, 3) fp32 CUDA weight (any strides) -> U (36, Cin, Cout), U[6a+b] = (G w G^T)[a,b]
    (csrc/trainops.hip, once per iteration and convolution).  adjoint=True: the weight of the
    input-gradient convolution, w^T rotated by 180 degrees -> (36, Cout, Cin)."""
    from . import _lib
    from .ops import _ptr, _stream
    if not (w.is_cuda and w.dtype == torch.float32 and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3)):
        raise TypeError('transform_weight needs a (Cout, Cin, 3, 3) fp32 CUDA weight')
    cout, cin = int(w.shape[0]), int(w.shape[1])
    so, si, sy, sx = w.stride()
    n_in, n_out, s_in, s_out = (cout, cin, so, si) if adjoint else (cin, cout, si, so)
    u = torch.empty((36, n_in, n_out), dtype=torch.float32, device=w.device)
    _lib.check(_lib.lib().ia_wino_weight_transform(_ptr(w), n_in, n_out, s_in, s_out, sy, sx,
                                                   int(bool(adjoint)), _G_HOST, _ptr(u), _stream()),
               'ia_wino_weight_transform')
    return u


def untransform_weight_grad(du, like=None):
    """(36, Cin, Cout) gradient w.r.t. U = G g G^T  ->  (Cout, Cin, 3, 3) gradient w.r.t. g, with the
    strides of `like` (the weight: contiguous or channels-last) when given"""
    from . import _lib
    from .ops import _ptr, _stream
    du = du.contiguous()
    cin, cout = int(du.shape[1]), int(du.shape[2])
    if like is not None and tuple(like.shape) == (cout, cin, 3, 3):
        dw = torch.empty_like(like, dtype=torch.float32)
    else:
        dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=du.device)
    so, si, sy, sx = dw.stride()
    _lib.check(_lib.lib().ia_wino_weight_grad(_ptr(du), cin, cout, _G_HOST, _ptr(dw), si, so, sy, sx,
                                              _stream()), 'ia_wino_weight_grad')
    return dw


def relu_bwd_bias_grad(dy, y=None, bias_grad=True):
    """channels-last (B, C, H, W) fp32: (dy masked by y > 0 [dy itself when y is None],
    per-channel sums of the masked gradient [None unless bias_grad]) in one pass"""
    from . import _lib
    from .ops import _ptr, _stream
    dy = _cl(dy)
    B, Cn, H, Wd = dy.shape
    if y is None and not bias_grad:
        return dy, None
    if Cn % 4:
        g = dy if y is None else torch.ops.aten.threshold_backward(dy, y, 0)
        return g, (g.sum((0, 2, 3)) if bias_grad else None)
    g = dy if y is None else torch.empty_like(dy)
    db = ws = None
    nbytes = 0
    if bias_grad:
        from .ops import _workspace
        db = torch.empty(Cn, dtype=torch.float32, device=dy.device)
        nbytes = int(_lib.lib().ia_relu_bwd_bias_grad_workspace_bytes(B * H * Wd, Cn))
        ws = _workspace(dy.device, nbytes)
    _lib.check(_lib.lib().ia_relu_bwd_bias_grad(_ptr(dy), None if y is None else _ptr(_cl(y)),
                                                B * H * Wd, Cn, None if y is None else _ptr(g),
                                                _ptr(db), _ptr(ws), nbytes, _stream()),
               'ia_relu_bwd_bias_grad')
    return g, db


DU = 'lt'             # 'lt': ops.gemm_tn (library candidates timed per shape); 'bmm': torch.bmm


class _WinoConvLevels(torch.autograd.Function):
    """conv3x3 (stride 1, pad 1) + bias (+ ReLU) with ONE weight over a list of level tensors"""

    @staticmethod
    def forward(ctx, weight, bias, relu, *xs):
        xs = [_cl(x) for x in xs]
        plan = _plan(xs)
        cout = weight.shape[0]
        with torch.no_grad():
            u = transform_weight(weight)                                     # (36, Cin, Cout)
            b = None if bias is None else bias.detach().float().contiguous()
            keep = ctx.needs_input_grad[0]
            out = _conv_levels(plan, xs, u, b, relu, cout, 'f', keep_v=keep)
            ys, v = out if keep else (out, None)
        ctx.relu, ctx.has_bias, ctx.plan, ctx.L = bool(relu), bias is not None, plan, len(xs)
        # V (the transformed input) replaces the input itself: it is what the Winograd-domain
        # weight gradient multiplies, and nothing else of x is needed in backward
        ctx.save_for_backward(weight, *(ys if relu else []), *([v] if keep else []))
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors
        L = ctx.L
        weight = saved[0]
        ys = list(saved[1:1 + L]) if ctx.relu else None
        v = saved[-1] if (len(saved) > 1 + (L if ctx.relu else 0)) else None
        plan = ctx.plan
        with torch.no_grad():
            want_b = ctx.has_bias and ctx.needs_input_grad[1]
            db = None
            masked = []
            for l, d in enumerate(dys):              # ReLU backward (dy where y > 0) + bias sums
                g, b = relu_bwd_bias_grad(d, ys[l] if ys is not None else None, want_b)
                masked.append(g)
                if want_b:
                    db = b if db is None else db + b
            dys = masked
            need_x = any(ctx.needs_input_grad[3 + l] for l in range(L))
            dxs = [None] * L
            if need_x:
                # adjoint of correlation with w (pad 1) = correlation with w^T flipped (pad 1)
                ut = transform_weight(weight, adjoint=True)                   # (36, Cout, Cin)
                dxs = _conv_levels(plan, dys, ut, None, False, weight.shape[1], 'b')
            dw = None
            if ctx.needs_input_grad[0] and v is not None:
                cout = weight.shape[0]
                dm = grad_output_transform(plan, dys, plan.buf('tdm', (36, plan.T, cout)))
                from .ops import gemm_tn
                du = gemm_tn(v, dm) if DU == 'lt' else torch.bmm(v.transpose(1, 2), dm)   # (36, Cin, Cout)
                dw = untransform_weight_grad(du, like=weight)
        return (dw, db, None) + tuple(dxs)


def wino_conv_levels(xs, weight, bias=None, relu=False):
    """xs: list of (B, Cin, H_l, W_l) fp32 CUDA tensors; -> list of (B, Cout, H_l, W_l)
    channels-last tensors.  Cin and Cout must be multiples of 4."""
    return list(_WinoConvLevels.apply(weight, bias, relu, *xs))


def _plain_3x3(conv):
    """3x3 / stride 1 / pad 1 / dilation 1 / groups 1 nn.Conv2d -- what the Winograd route computes"""
    return (type(conv) is torch.nn.Conv2d and tuple(conv.kernel_size) == (3, 3)
            and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.bias is not None
            and conv.padding_mode == 'zeros')


def _library_loads():
    from . import _lib
    try:
        _lib.lib()
        return True
    except Exception:                     # missing .so: the plain modules still train (MIOpen)
        return False


def usable(feats, head):
    towers = list(head.cls_convs) + list(head.reg_convs)
    return (torch.is_grad_enabled() and all(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                                            for x in feats)
            and head.in_channels % 4 == 0 and head.feat_channels % 4 == 0
            and (head.num_anchors * head.cls_out_channels) % 4 == 0
            and all(not m.with_norm and m.with_activatation and _plain_3x3(m.conv) for m in towers)
            and all(_plain_3x3(c) for c in (head.retina_cls, head.retina_reg, head.retina_iou))
            and _library_loads())


def head_forward(head, feats):
    """IoUawareRetinaHead.forward (multi_apply(forward_single), reference :171-219) with every
    convolution evaluated for all levels at once.  Returns (cls[L], reg[L], iou[L]); reg / iou are
    channel slices of one 48-channel output (retina_reg | retina_iou | zero padding)."""
    cls_feat = reg_feat = [_cl(x) for x in feats]
    for conv in head.cls_convs:
        cls_feat = wino_conv_levels(cls_feat, conv.conv.weight, conv.conv.bias, relu=True)
    for conv in head.reg_convs:
        reg_feat = wino_conv_levels(reg_feat, conv.conv.weight, conv.conv.bias, relu=True)
    cls = wino_conv_levels(cls_feat, head.retina_cls.weight, head.retina_cls.bias)
    n_reg, n_iou = head.retina_reg.out_channels, head.retina_iou.out_channels
    pad = (-(n_reg + n_iou)) % 4
    w_ri = torch.cat([head.retina_reg.weight, head.retina_iou.weight] +
                     ([head.retina_reg.weight.new_zeros((pad,) + tuple(head.retina_reg.weight.shape[1:]))]
                      if pad else []))
    b_ri = torch.cat([head.retina_reg.bias, head.retina_iou.bias] +
                     ([head.retina_reg.bias.new_zeros(pad)] if pad else []))
    ri = wino_conv_levels(reg_feat, w_ri, b_ri)
    reg = [t[:, :n_reg] for t in ri]
    iou = [t[:, n_reg:n_reg + n_iou] for t in ri]
    return cls, reg, iou
