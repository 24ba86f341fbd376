"""Winograd F(4x4,3x3) inference path for the 3x3 / stride-1 convolutions of the IoU-aware
RetinaNet head (reference iou_aware_retina_head.py:171-219: cls / reg towers, retina_cls,
retina_reg, retina_iou) and of the FPN output convolutions (fpn.py:124-127).

    activations (channels-last, all five levels)  --k_wino_in-->   V  (36 matrices tiles x Cin)
    V . U   (36 [x2 towers] plain fp32 GEMMs: hipBLASLt, strided batched, csrc/gemm.hip)    -->  M
    M  --k_wino_out (+bias, +ReLU)-->  next activations / the head outputs, channels-last

The head's weights are shared by the pyramid levels, so every layer is ONE batched GEMM over the
tiles of all levels; the two towers run side by side (first layer: one GEMM with 512 output
columns on the shared input, later layers: 72 matrices).  36 multiplications per 4x4 output tile
instead of 144: these convolutions are 63 % of the network's multiply-adds.

Numerics: fp32 throughout; the Winograd transforms reassociate the sums, outputs agree with a
direct convolution to ~1e-5 of the activation scale (tests/test_gpu_winograd.py).  The weights are
transformed once (fp64 -> fp32) when the runner is built: build it AFTER loading a checkpoint.
"""
import collections
import ctypes as C

import numpy as np
import torch

from . import _lib
from .ops import _ptr, _stream, stream_id

_G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
               [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)


_G_DEV = {}


def transform_weight(w):
    """(Cout, Cin, 3, 3) conv weight -> U (36, Cin, Cout) fp32, U[6i+j] = (G g G^T)[i, j]."""
    if w.dim() != 4 or w.shape[2] != 3 or w.shape[3] != 3:
        raise ValueError('Winograd F(4,3) needs a 3x3 kernel, got %s' % (tuple(w.shape),))
    g = w.detach().to(torch.float64)
    G = _G_DEV.get(g.device)
    if G is None:
        G = _G_DEV[g.device] = torch.from_numpy(_G).to(g.device)
    u = torch.einsum('ik,ockl,jl->ijco', G, g, G)           # (6, 6, Cin, Cout)
    return u.reshape(36, g.shape[1], g.shape[0]).to(torch.float32).contiguous()


def _wino_geom(sizes, batch):
    g = _lib.WinoGeom()
    g.num_levels, g.batch = len(sizes), int(batch)
    for l, (h, w) in enumerate(sizes):
        g.H[l], g.W[l] = int(h), int(w)
    tiles = C.c_int32()
    _lib.check(_lib.lib().ia_wino_tiles(C.byref(g), C.byref(tiles)), 'ia_wino_tiles')
    return g, tiles.value


def _usable(x):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 4 == 0
            and x.is_contiguous(memory_format=torch.channels_last))


_SCRATCH = {}


def _scratch(device, name, shape):
    """V / M staging buffers are shared by every Winograd layer of the process: the layers run
    one after the other on one stream, so a buffer is free again when the next layer starts."""
    n = 1
    for d in shape:
        n *= int(d)
    key = (device, stream_id(), name)
    b = _SCRATCH.get(key)
    if b is None or b.numel() < n:
        b = _SCRATCH[key] = torch.empty(n, dtype=torch.float32, device=device)
    return b[:n].view(*shape)


_PLAN_ENTRIES = 4     # plans kept per layer object: the head's plan owns two ping-pong activation sets
                      # (2 x B x 2F x sum(HW) floats: 92 MB at batch 1, 736 MB at batch 8 of 800 x 1344)


def _plan_for(plans, key, make):
    """least-recently-used cache of _Plan objects: evaluation meets many pad shapes (keep-ratio
    resize, padded to /32), and a plan per shape kept for the life of the model pinned its
    activation buffers for good (the growth ADVICE r3 found in the bf16 head's buffers)"""
    plan = plans.get(key)
    if plan is None:
        while len(plans) >= _PLAN_ENTRIES:
            plans.popitem(last=False)
        plan = plans[key] = make()
    else:
        plans.move_to_end(key)
    return plan


class _Plan(object):
    """buffers of one (batch, feature-map sizes) configuration"""

    def __init__(self, sizes, batch, device):
        self.sizes, self.batch = list(sizes), batch
        self.geom, self.T = _wino_geom(sizes, batch)
        self.device = device
        self._bufs = {}

    def buf(self, name, shape):
        return _scratch(self.device, name, shape)

    def acts(self, name, channels):
        """per-level channels-last activation tensors (B, C, H, W)"""
        key = ('acts', name, channels)
        a = self._bufs.get(key)
        if a is None:
            a = [torch.empty((self.batch, channels, h, w), dtype=torch.float32, device=self.device,
                             memory_format=torch.channels_last) for (h, w) in self.sizes]
            self._bufs[key] = a
        return a


# Measurement hook (bench.py): when TIMING is a list, every transform launch appends
# (kind, start event, end event, algorithmic bytes) -- HIP events on the launch stream.  Bytes:
# input transform = the activation read once (16 pixels per tile) + V written (36 per tile);
# output transform = M read (36 per tile) + the activation written (16 per tile), fp32.
TIMING = None


def _timed(kind, nbytes, launch):
    if TIMING is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    TIMING.append((kind, e0, e1, nbytes))


def input_transform(plan, xs, groups, out, pre=None):
    """pre = (scale or None, shift, relu): the input is read as relu?(x * scale + shift)"""
    ptrs = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    ps, pb, pr = pre if pre is not None else (None, None, False)
    ch = int(xs[0].shape[1])
    _timed('in', plan.T * ch * 4 * (16 + 36), lambda: _lib.check(
        _lib.lib().ia_wino_input_transform(C.byref(plan.geom), ptrs, ch, int(groups), _ptr(ps),
                                           _ptr(pb), int(bool(pr)), _ptr(out), _stream()),
        'ia_wino_input_transform'))
    return out


def output_transform(plan, m, channels, groups, bias, relu, segments):
    """segments: list of (c0, n, dst_tensors, dst_offset)"""
    segs = (_lib.WinoSeg * len(segments))()
    for k, (c0, n, dst, off) in enumerate(segments):
        segs[k].c0, segs[k].n = int(c0), int(n)
        segs[k].dst_channels, segs[k].dst_offset = int(dst[0].shape[1]), int(off)
        for l, t in enumerate(dst):
            segs[k].dst[l] = t.data_ptr()
    _timed('out', plan.T * int(channels) * 4 * (36 + 16), lambda: _lib.check(
        _lib.lib().ia_wino_output_transform(C.byref(plan.geom), _ptr(m), int(channels), int(groups),
                                            _ptr(bias), int(bool(relu)), len(segments), segs,
                                            _stream()), 'ia_wino_output_transform'))


_LT_WS_BYTES = 128 << 20


STREAM_BMM = True
_STREAM_BMM_SHAPES = ((64, 64),)      # (128, 128) and (256, 48) measured slower than the library: 137 / 148 vs 124 / 139 us


def batched_gemm(v, u, out):
    """out[b] = v[b] @ u[b] through hipBLASLt (csrc/gemm.hip); the kernel of a shape comes from the
    committed tuning table (ops.gemm_table_load: found offline by timing the library's candidates,
    worth ~20 % over the heuristic's pick at these shapes), never from a timing race at run time"""
    from . import ops
    ops._ensure_gemm_table()
    batch, rows, k = v.shape
    n = u.shape[2]
    if not (v.is_contiguous() and u.is_contiguous() and out.is_contiguous()):
        raise ValueError('batched_gemm needs contiguous stacks')
    if STREAM_BMM and (k, n) in _STREAM_BMM_SHAPES and rows >= 4096 and v.dtype == torch.float32:
        # HBM-bound products (<= 16 K weights per matrix): this library's streaming MFMA kernel,
        # weights in LDS -- the library's kernels reach 3.4 TB/s on these shapes
        _lib.check(_lib.lib().ia_batched_gemm_stream(_ptr(v), _ptr(u), _ptr(out), int(batch), int(rows),
                                                     int(k), int(n), _stream()), 'ia_batched_gemm_stream')
        return out
    ws = _scratch(v.device, 'lt_ws', (_LT_WS_BYTES // 4,))
    _lib.check(_lib.lib().ia_batched_gemm(_ptr(v), _ptr(u), _ptr(out), int(batch), int(rows), int(k),
                                          int(n), _ptr(ws), _LT_WS_BYTES, _stream()),
               'ia_batched_gemm')
    return out


class WinogradConv3x3(object):
    """one 3x3 / stride-1 / pad-1 convolution (+bias, +ReLU) over a list of channels-last level
    tensors that do NOT share the weight with other layers' inputs (FPN output convs: one
    instance per level)."""

    def __init__(self, weight, bias, relu=False):
        self.u = transform_weight(weight)
        self.bias = None if bias is None else bias.detach().to(torch.float32).contiguous()
        self.relu = relu
        self.cin, self.cout = self.u.shape[1], self.u.shape[2]
        if self.cout % 4:
            raise ValueError('output channels must be a multiple of 4')
        self._plans = collections.OrderedDict()

    def usable(self, x):
        # raw kernels, no autograd graph: grad mode off, or an input that carries no gradient
        # (a frozen stage during training; fuse._fast has checked the parameters)
        return _usable(x) and x.shape[1] == self.cin \
            and not (torch.is_grad_enabled() and x.requires_grad)

    def __call__(self, x, pre=None):
        """pre = (scale, shift, relu): x is the raw output of the convolution in front, its folded
        BatchNorm / ReLU is applied while the input transform loads it"""
        # the plan owns scratch buffers (V, M): one per stream, so forwards on different streams
        # do not share them
        key = (x.shape[0], tuple(x.shape[-2:]), x.device, stream_id())
        plan = _plan_for(self._plans, key, lambda: _Plan([tuple(x.shape[-2:])], x.shape[0], x.device))
        v = input_transform(plan, [x], 1, plan.buf('v', (36, plan.T, self.cin)), pre)
        m = batched_gemm(v, self.u, plan.buf('m', (36, plan.T, self.cout)))
        y = torch.empty((x.shape[0], self.cout) + tuple(x.shape[-2:]), dtype=torch.float32,
                        device=x.device, memory_format=torch.channels_last)
        output_transform(plan, m, self.cout, 1, self.bias, self.relu, [(0, self.cout, [y], 0)])
        return y


class WinogradHead(object):
    """the conv towers and output convolutions of an IoUawareRetinaHead, all levels at once."""

    def __init__(self, head):
        convs_c, convs_r = list(head.cls_convs), list(head.reg_convs)
        if any(m.with_norm or not m.with_activatation for m in convs_c + convs_r):
            raise NotImplementedError('towers with norm layers / without ReLU')
        self.n_layers = len(convs_c)
        F = head.feat_channels
        if head.in_channels % 4 or F % 4:
            raise ValueError('channel counts must be multiples of 4')
        self.F, self.cin = F, head.in_channels

        def wb(m):
            conv = m.conv
            b = conv.bias if conv.bias is not None else torch.zeros(conv.out_channels,
                                                                     device=conv.weight.device)
            return transform_weight(conv.weight), b.detach().float()

        # layer 0: both towers read the FPN feature -> one GEMM with 2F output columns
        (uc, bc), (ur, br) = wb(convs_c[0]), wb(convs_r[0])
        self.u0 = torch.cat([uc, ur], dim=2).contiguous()               # (36, Cin, 2F)
        self.b0 = torch.cat([bc, br]).contiguous()
        # layers 1..: two groups side by side -> 72 matrices
        self.u, self.b = [], []
        for i in range(1, self.n_layers):
            (uc, bc), (ur, br) = wb(convs_c[i]), wb(convs_r[i])
            self.u.append(torch.cat([uc, ur], dim=0).contiguous())       # (72, F, F)
            self.b.append(torch.cat([bc, br]).contiguous())
        # outputs: retina_cls on the cls tower; retina_reg | retina_iou on the reg tower
        self.c_cls = head.retina_cls.out_channels
        self.c_reg, self.c_iou = head.retina_reg.out_channels, head.retina_iou.out_channels
        if self.c_cls % 4:
            raise ValueError('A*C must be a multiple of 4')
        self.u_cls = transform_weight(head.retina_cls.weight)
        self.b_cls = head.retina_cls.bias.detach().float().contiguous()
        n_ri = self.c_reg + self.c_iou
        self.n_ri_pad = (n_ri + 15) // 16 * 16
        u_ri = torch.zeros((36, F, self.n_ri_pad), dtype=torch.float32, device=self.u_cls.device)
        u_ri[:, :, :self.c_reg] = transform_weight(head.retina_reg.weight)
        u_ri[:, :, self.c_reg:n_ri] = transform_weight(head.retina_iou.weight)
        self.u_ri = u_ri.contiguous()
        b_ri = torch.zeros(self.n_ri_pad, dtype=torch.float32, device=self.u_cls.device)
        b_ri[:self.c_reg] = head.retina_reg.bias.detach().float()
        b_ri[self.c_reg:n_ri] = head.retina_iou.bias.detach().float()
        self.b_ri = b_ri.contiguous()
        self._plans = collections.OrderedDict()

    def usable(self, feats):
        return all(_usable(x) and x.shape[1] == self.cin for x in feats) \
            and not torch.is_grad_enabled()

    def __call__(self, feats):
        """feats: per-level (B, Cin, H, W) channels-last fp32 -> (cls[L], reg[L], iou[L])"""
        B = feats[0].shape[0]
        sizes = [tuple(x.shape[-2:]) for x in feats]
        key = (B, tuple(sizes), feats[0].device, stream_id())
        plan = _plan_for(self._plans, key, lambda: _Plan(sizes, B, feats[0].device))
        T, F = plan.T, self.F
        # layer 0
        v = input_transform(plan, feats, 1, plan.buf('v', (36, T, self.cin)))
        m = batched_gemm(v, self.u0, plan.buf('m', (36, T, 2 * F)))
        acts = plan.acts('a', 2 * F)
        output_transform(plan, m, 2 * F, 1, self.b0, True, [(0, 2 * F, acts, 0)])
        # layers 1..n-1: groups = 2 (cls tower = channels [0,F), reg tower = [F,2F))
        for u, b in zip(self.u, self.b):
            v = input_transform(plan, acts, 2, plan.buf('v', (72, T, F)))
            m = batched_gemm(v, u, plan.buf('m', (72, T, F)))
            nxt = plan.acts('b' if acts is plan.acts('a', 2 * F) else 'a', 2 * F)
            output_transform(plan, m, 2 * F, 2, b, True, [(0, 2 * F, nxt, 0)])
            acts = nxt
        # outputs
        v = input_transform(plan, acts, 2, plan.buf('v', (72, T, F)))
        new = lambda c: [torch.empty((B, c, h, w), dtype=torch.float32, device=feats[0].device,  # noqa: E731
                                     memory_format=torch.channels_last) for (h, w) in sizes]
        cls, reg, iou = new(self.c_cls), new(self.c_reg), new(self.c_iou)
        # the 548 MB class logits are written FIRST: the (MFMA-bound) reg / iou GEMM behind them
        # gives their write-back time to drain before the row-max kernel streams them back in
        # (round 6, measured again in bench steps, 2 x 20 steps each: class logits LAST -> decode stage 0.133 ms
        # in-step against 0.120-0.121 this way; gpurun_out/r06_ab_clslast.txt)
        m_cls = batched_gemm(v[:36], self.u_cls, plan.buf('mc', (36, T, self.c_cls)))
        output_transform(plan, m_cls, self.c_cls, 1, self.b_cls, False, [(0, self.c_cls, cls, 0)])
        m_ri = batched_gemm(v[36:], self.u_ri, plan.buf('mr', (36, T, self.n_ri_pad)))
        output_transform(plan, m_ri, self.n_ri_pad, 1, self.b_ri, False,
                         [(0, self.c_reg, reg, 0), (self.c_reg, self.c_iou, iou, 0)])
        return cls, reg, iou
