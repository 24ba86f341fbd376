"""torch-tensor front-end of the C-ABI kernels.

PyTorch is plumbing here (device memory, streams); every function below
launches hand-written gfx950 kernels through libiouaware_hip.so on the current
torch stream and returns device tensors.  No function has a CPU path: tensors
must live on a ROCm device.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import HeadGeom, LevelPtrs, IA_F32, IA_BF16


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_id():
    """integer handle of the current stream of the current device (cache keys)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    """the current stream of the current device as a hipStream_t.  torch.cuda.current_stream() builds a
    Stream object through four layers of Python (16 us per call under the profiler, three calls per loss
    evaluation, one per C-ABI call on the inference path); the raw-handle accessor behind it costs a
    fraction of a microsecond."""
    return C.c_void_p(stream_id())


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dtype_code(t):
    if t.dtype == torch.float32:
        return IA_F32
    if t.dtype == torch.bfloat16:
        return IA_BF16
    raise TypeError('head outputs must be float32 or bfloat16, got %s' % t.dtype)


def _require_gpu(t, name):
    if not t.is_cuda:
        raise _lib.IouAwareLibraryError(
            '%s is on %s: the IoU-aware head kernels are gfx950 HIP kernels and have no CPU '
            'fallback' % (name, t.device))


class HeadGeometry(object):
    """Static geometry of an anchor head for one set of feature-map sizes
    (fills the C struct ia_head_geom)."""

    def __init__(self, featmap_sizes, strides, base_anchors, num_classes, nms_pre=-1,
                 means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), softmax=False):
        """num_classes: foreground classes C (the score columns).  softmax=True: the head's
        use_sigmoid_cls=False branch (iou_aware_retina_head.py:506-507,540-541): the class tensors
        carry A * (C + 1) channels, channel 0 of an anchor = background."""
        base = np.asarray(base_anchors, dtype=np.float32)
        L, A = base.shape[0], base.shape[1]
        if L != len(featmap_sizes) or L != len(strides):
            raise ValueError('level count mismatch')
        if L > _lib.IA_MAX_LEVELS or A > _lib.IA_MAX_ANCHORS:
            raise ValueError('unsupported head geometry (levels %d, anchors %d)' % (L, A))
        g = HeadGeom()
        g.num_levels, g.num_anchors, g.num_classes, g.nms_pre = L, A, int(num_classes), int(nms_pre)
        for l, ((h, w), s) in enumerate(zip(featmap_sizes, strides)):
            g.H[l], g.W[l], g.stride[l] = int(h), int(w), int(s)
            for a in range(A):
                for k in range(4):
                    g.base_anchors[l][a][k] = float(base[l, a, k])
        for k in range(4):
            g.means[k], g.stds[k] = float(means[k]), float(stds[k])
        g.cls_activation = _lib.IA_CLS_SOFTMAX if softmax else _lib.IA_CLS_SIGMOID
        self.struct = g
        self.L, self.A, self.C = L, A, int(num_classes)
        self.softmax = bool(softmax)
        self.Cin = self.C + 1 if softmax else self.C          # class channels per anchor
        self.featmap_sizes = [tuple(int(v) for v in s) for s in featmap_sizes]
        self.strides = [int(s) for s in strides]
        n, r, rs = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().ia_geom_sizes(C.byref(g), C.byref(n), C.byref(r), C.byref(rs)),
                   'ia_geom_sizes')
        self.N, self.R, self.Rs = n.value, r.value, rs.value
        self.level_anchors = [h * w * A for (h, w) in self.featmap_sizes]
        self.level_cands = [min(nms_pre, n_) if nms_pre > 0 else n_ for n_ in self.level_anchors]
        self.layout = _lib.IA_LAYOUT_NCHW
        self._twin = None
        # what the workspace carve-up depends on (besides batch, layout and dtype)
        self.key = (tuple(self.featmap_sizes), A, int(num_classes), int(nms_pre), bool(softmax))

    def ref(self):
        return C.byref(self.struct)

    def with_layout(self, layout):
        """the same geometry for head outputs stored in the other memory order"""
        if layout == self.layout:
            return self
        if self._twin is None:
            import copy
            t = copy.copy(self)
            t.struct = HeadGeom.from_buffer_copy(self.struct)
            t.struct.layout = layout
            t.layout, t._twin = layout, self
            self._twin = t
        return self._twin


def to_nchw(t):
    """Contiguous NCHW view/copy of a 4-D tensor; channels-last inputs go through the
    LDS-tiled HIP transpose instead of torch's generic strided copy."""
    if t.is_contiguous():
        return t
    if t.dim() == 4 and t.is_cuda and t.is_contiguous(memory_format=torch.channels_last) \
            and t.dtype in (torch.float32, torch.bfloat16):
        out = torch.empty(t.shape, dtype=t.dtype, device=t.device)
        n, c, h, w = t.shape
        _lib.check(_lib.lib().ia_nhwc_to_nchw(_ptr(t), _ptr(out), _dtype_code(t), n, c, h * w,
                                              _stream()), 'ia_nhwc_to_nchw')
        return out
    return t.contiguous()


def _nhwc_ok(geom, tensors):
    """channels-last head outputs are consumed in place (no transposes) when every tensor is
    channels-last contiguous and a class row is a whole number (<= 32) of 16-byte vectors"""
    row = geom.Cin * tensors[0].element_size()
    if row % 16 != 0 or row > 512:
        return False
    if all(t.is_contiguous() for t in tensors):
        return False
    return all(t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)
               for t in tensors)


def level_ptrs(geom, cls, reg, iou):
    """Validate the per-level head outputs and pack their device pointers.
    -> (ptrs, batch, dtype code, geometry for the memory order the tensors are in)"""
    if not (len(cls) == len(reg) == len(iou) == geom.L):
        raise AssertionError('expected %d levels' % geom.L)
    p = LevelPtrs()
    B = cls[0].shape[0]
    dt = _dtype_code(cls[0])
    nhwc = _nhwc_ok(geom, list(cls) + list(reg) + list(iou))
    geom = geom.with_layout(_lib.IA_LAYOUT_NHWC if nhwc else _lib.IA_LAYOUT_NCHW)
    for l in range(geom.L):
        h, w = geom.featmap_sizes[l]
        for name, t, ch in (('cls_score', cls[l], geom.A * geom.Cin), ('bbox_pred', reg[l], geom.A * 4),
                            ('iou_pred', iou[l], geom.A)):
            _require_gpu(t, name)
            if tuple(t.shape) != (B, ch, h, w):
                raise AssertionError('%s level %d has shape %s, expected %s'
                                     % (name, l, tuple(t.shape), (B, ch, h, w)))
            if _dtype_code(t) != dt:
                raise TypeError('mixed dtypes in head outputs')
        if not nhwc:
            cls[l] = to_nchw(cls[l])
            reg[l] = to_nchw(reg[l])
            iou[l] = to_nchw(iou[l])
        p.cls[l], p.reg[l], p.iou[l] = cls[l].data_ptr(), reg[l].data_ptr(), iou[l].data_ptr()
    return p, B, dt, geom


_ws_cache = {}


def _own_workspace(device, nbytes):
    """a workspace that one autograd node keeps from forward to backward (never shared)"""
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def _workspace(device, nbytes):
    key = (device.index, stream_id())
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


import collections
_state_ws_cache = collections.OrderedDict()
_STATE_WS_ENTRIES = 64          # (pad shape, batch) combinations kept per process; ~7 MB per image each


def _state_workspace(device, nbytes, layout_key):
    """The workspace of ia_get_bboxes / ia_decode_stage carries state between calls (the flag words
    of the fused row-max + filter launch: include/iouaware.h, WORKSPACE CONTRACT): its own buffer
    per (device, stream, geometry, batch) -- the carve-up depends on those --, zero-filled when
    created, shared with nothing else."""
    key = (device.index, stream_id(), layout_key, int(nbytes))
    ws = _state_ws_cache.get(key)
    if ws is None:
        while len(_state_ws_cache) >= _STATE_WS_ENTRIES:       # least recently used first
            _state_ws_cache.popitem(last=False)
        ws = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
        _state_ws_cache[key] = ws
    else:
        _state_ws_cache.move_to_end(key)
    return ws


def get_bboxes_status(geom, batch, ws):
    """(id of the last call whose fused launch timed out and fell back to the dense selection,
    number of such calls) from the status words in `ws`; (0, 0) = never.  A host synchronisation:
    telemetry and tests, not the product path (the fallback's result is correct by itself)."""
    off = _lib.lib().ia_get_bboxes_status_offset(geom.ref(), int(batch))
    w = ws[off:off + 8].view(torch.int32).cpu()
    return int(w[0]) & 0xffffffff, int(w[1]) & 0xffffffff


def fused_spin_limit(limit=-1):
    """bound of the fused row-max + filter launch's waits (tests: 0 forces the fallback path of
    k_sel_final; -1 restores the default)"""
    _lib.check(_lib.lib().ia_debug_fused_spin_limit(int(limit)), 'ia_debug_fused_spin_limit')


def stage_events(begin=None, end=None):
    """ia_profile_stage_events: the decode stage of every later get_bboxes / DecodeStage call records
    these two torch.cuda.Event(enable_timing=True) objects on its stream (in front of the row-max
    launch, behind the gather); None, None switches the hook off.  The events must have been
    recorded once (torch creates the HIP event lazily) and stay alive while they are set."""
    if begin is None and end is None:
        _lib.check(_lib.lib().ia_profile_stage_events(None, None), 'ia_profile_stage_events')
        return
    if not (begin.cuda_event and end.cuda_event):
        raise ValueError('record the events once before handing them over (torch creates them lazily)')
    _lib.check(_lib.lib().ia_profile_stage_events(C.c_void_p(begin.cuda_event),
                                                  C.c_void_p(end.cuda_event)),
               'ia_profile_stage_events')


def state_workspace_for(geom, cls, reg, iou):
    """the persistent workspace `get_bboxes` uses for these head outputs (tests / telemetry)"""
    p, B, dt, g = level_ptrs(geom, list(cls), list(reg), list(iou))
    nbytes = _lib.lib().ia_get_bboxes_workspace_bytes(g.ref(), B)
    return g, B, _state_workspace(cls[0].device, nbytes, (g.key, g.layout, B, dt))


_meta_cache = {}


def _meta_tensors(img_shapes, scale_factors, device):
    """(B,2) image sizes and (B,4) scale factors on the device.  Cached by value: a detector sees
    the same few (img_shape, scale_factor) combinations over and over, the upload happens once
    per combination -- two host-to-device copies less per call, and none at all while a step is
    being captured into a HIP graph (a captured copy from a temporary host tensor would read
    freed memory on replay)."""
    sf = []
    for s in scale_factors:
        v = np.asarray(s, dtype=np.float32).reshape(-1)
        sf.append(np.repeat(v, 4) if v.size == 1 else v)
    sf = np.stack(sf).astype(np.float32)
    hw = np.asarray([[float(s[0]), float(s[1])] for s in img_shapes], np.float32)
    key = (str(device), hw.tobytes(), sf.tobytes())
    hit = _meta_cache.get(key)
    if hit is None:
        if len(_meta_cache) >= 256:
            _meta_cache.clear()
        hit = _meta_cache[key] = (torch.from_numpy(hw).to(device), torch.from_numpy(sf).to(device))
    return hit


def get_bboxes(geom, cls, reg, iou, img_shapes, scale_factors, rescale, score_thr, iou_thr,
               max_per_img, debug=False, soft=None, lazy=True, lazy_candidates=0):
    """Whole post-conv inference path for a batch.

    soft: None for hard NMS (one C-ABI call), or dict(method=, sigma=, min_score=) for
    test_cfg.nms.type='soft_nms' (the stage calls with ia_multiclass_soft_nms at the end).
    lazy: evaluate the NMS lazily (ia_get_bboxes_lazy: same detections, the class problems are
    not resolved completely); debug=True always takes the complete path, whose per-class keep
    lists are part of the debug views.

    Returns device tensors dets (B,max_per_img,5) f32, labels (B,max_per_img) i32,
    rows (B,max_per_img) i32 (candidate row ids), num (B) i32.  With debug=True
    also returns the workspace views (rowmax, cand_idx, boxes, scores_t,
    keep_count, keep_rows) for stage-level parity tests.
    """
    cls, reg, iou = list(cls), list(reg), list(iou)
    if soft is not None:
        geom = geometry_for(geom, cls, reg, iou)
        if geom.R > _lib.IA_MAX_CANDIDATES or max_per_img > _lib.IA_MAX_PER_IMG:
            return _get_bboxes_per_class(geom, cls, reg, iou, img_shapes, scale_factors, rescale, score_thr,
                                         iou_thr, max_per_img, debug, soft=soft)
        cand = select_topk(geom, decode_fuse_rowmax(geom, cls, reg, iou))
        boxes, scores_t, _ = gather_decode(geom, cls, reg, iou, cand, img_shapes, scale_factors,
                                           rescale)
        out = multiclass_soft_nms(boxes, scores_t, geom.R, score_thr, iou_thr, max_per_img, **soft)
        if not debug:
            return out[:4]
        return out[:4] + (dict(cand_idx=cand, boxes=boxes, scores_t=scores_t, keep_count=out[4],
                               keep_rows=out[5]),)
    p, B, dt, geom = level_ptrs(geom, cls, reg, iou)
    dev = cls[0].device
    L = _lib.lib()
    nbytes = L.ia_get_bboxes_workspace_bytes(geom.ref(), B)
    if (nbytes == 0 and geom.R > _lib.IA_MAX_CANDIDATES) or max_per_img > _lib.IA_MAX_PER_IMG:
        # beyond the batched entry's capacities -- the reference has none (bbox_nms.py:33-56 takes
        # any number of candidates and any max_num): the stage entries, then one NMS per class
        return _get_bboxes_per_class(geom, cls, reg, iou, img_shapes, scale_factors, rescale, score_thr,
                                     iou_thr, max_per_img, debug)
    if nbytes == 0:
        raise _lib.IouAwareLibraryError('unsupported geometry / batch for ia_get_bboxes')
    ws = _state_workspace(dev, nbytes, (geom.key, geom.layout, B, dt))
    hw, sf = _meta_tensors(img_shapes, scale_factors, dev)
    dets = torch.empty((B, max_per_img, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((B, max_per_img), dtype=torch.int32, device=dev)
    rows = torch.empty((B, max_per_img), dtype=torch.int32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    if lazy and not debug:
        rc = L.ia_get_bboxes_lazy(geom.ref(), C.byref(p), B, dt, _ptr(hw), _ptr(sf),
                                  int(bool(rescale)), float(score_thr), float(iou_thr),
                                  int(max_per_img), int(lazy_candidates), _ptr(ws), nbytes,
                                  _ptr(dets), _ptr(labels), _ptr(rows), _ptr(num), _stream())
    else:
        rc = L.ia_get_bboxes(geom.ref(), C.byref(p), B, dt, _ptr(hw), _ptr(sf), int(bool(rescale)),
                             float(score_thr), float(iou_thr), int(max_per_img), _ptr(ws), nbytes,
                             _ptr(dets), _ptr(labels), _ptr(rows), _ptr(num), _stream())
    _lib.check(rc, 'ia_get_bboxes')
    if not debug:
        return dets, labels, rows, num
    off = (C.c_size_t * 8)()
    _lib.check(L.ia_get_bboxes_workspace_layout(geom.ref(), B, C.byref(off)), 'workspace_layout')

    def view(i, dtype, shape):
        n = int(np.prod(shape))
        return ws[off[i]:off[i] + n * 4].view(dtype).view(*shape)
    dbg = dict(rowmax=view(0, torch.float32, (B, geom.N)),
               cand_idx=view(1, torch.int32, (B, geom.R)),
               boxes=view(2, torch.float32, (B, geom.R, 4)),
               scores_t=view(3, torch.float32, (B, geom.C, geom.Rs)),
               keep_count=view(4, torch.int32, (B, geom.C)),
               keep_rows=view(5, torch.int32, (B, geom.C, geom.Rs)),
               fused_fallbacks=torch.tensor(get_bboxes_status(geom, B, ws)[1]))
    return dets, labels, rows, num, dbg


def _get_bboxes_per_class(geom, cls, reg, iou, img_shapes, scale_factors, rescale, score_thr, iou_thr,
                          max_per_img, debug=False, soft=None):
    """get_bboxes beyond the capacities of the batched C-ABI entry (more than IA_MAX_CANDIDATES
    candidates per image -- e.g. nms_pre = 2000 on five large levels -- or max_per_img above
    IA_MAX_PER_IMG): the decode stage through its stage entries (row-max, exact top-k, gather /
    decode: pure geometry, any size), then multiclass_nms AS THE REFERENCE WRITES IT
    (mmdet/core/post_processing/bbox_nms.py:33-56): one NMS per class on the single-problem entry
    (ia_nms takes any n), class-major concatenation, and the score sort only when more than
    max_per_img survive (stable: equal scores keep their concatenation order, this build's canonical
    order).  soft = dict(method=, sigma=, min_score=): soft-NMS per class instead (detections carry the
    decayed scores, a class's survivors in selection order; up to IA_MAX_CANDIDATES boxes per class
    problem).  Same return values as get_bboxes; a host synchronisation per class problem -- the
    slow, unbounded route."""
    geom = geometry_for(geom, cls, reg, iou)
    rowmax = decode_fuse_rowmax(geom, cls, reg, iou)
    cand = select_topk(geom, rowmax)
    boxes, scores_t, _ = gather_decode(geom, cls, reg, iou, cand, img_shapes, scale_factors, rescale)
    B, R, Cn, dev = boxes.shape[0], geom.R, geom.C, boxes.device
    dets = torch.zeros((B, max_per_img, 5), dtype=torch.float32, device=dev)
    labels = torch.full((B, max_per_img), -1, dtype=torch.int32, device=dev)
    rows = torch.full((B, max_per_img), -1, dtype=torch.int32, device=dev)
    num = torch.zeros((B,), dtype=torch.int32, device=dev)
    kc = torch.zeros((B, Cn), dtype=torch.int32, device=dev)
    kr = torch.zeros((B, Cn, geom.Rs), dtype=torch.int32, device=dev) if debug else None
    for b in range(B):
        over = scores_t[b, :, :R] > score_thr                       # (C, R)
        live = over.any(dim=1).nonzero().flatten().tolist()         # classes with a survivor of the threshold
        d_all, l_all, r_all = [], [], []
        for c in live:
            inds = over[c].nonzero().flatten()                      # ascending candidate rows
            d = torch.cat([boxes[b, inds], scores_t[b, c, inds, None]], dim=1)
            if soft is not None:
                kept, keep = soft_nms_dets(d, iou_thr, **soft)      # selection order, decayed scores
            else:
                keep = nms_indices(d, iou_thr)                      # ascending, like nms_cpu.cpp:58
                kept = d[keep]
            d_all.append(kept); r_all.append(inds[keep])
            l_all.append(torch.full((keep.numel(),), c, dtype=torch.int32, device=dev))
            kc[b, c] = keep.numel()
            if debug:
                kr[b, c, :keep.numel()] = inds[keep].to(torch.int32)
        if not d_all:
            continue
        d_all, l_all, r_all = torch.cat(d_all), torch.cat(l_all), torch.cat(r_all)
        if d_all.shape[0] > max_per_img:
            order = torch.sort(d_all[:, 4], descending=True, stable=True)[1][:max_per_img]
            d_all, l_all, r_all = d_all[order], l_all[order], r_all[order]
        n = d_all.shape[0]
        dets[b, :n], labels[b, :n], rows[b, :n], num[b] = d_all, l_all, r_all.to(torch.int32), n
    if not debug:
        return dets, labels, rows, num
    return dets, labels, rows, num, dict(rowmax=rowmax, cand_idx=cand, boxes=boxes, scores_t=scores_t,
                                         keep_count=kc, keep_rows=kr, fused_fallbacks=torch.tensor(0))


class DecodeStage(object):
    """SURVEY 8(d)'s decode stage (row-max -> top-k -> gather / decode) as ONE C-ABI call into a
    workspace allocated once: what ia_get_bboxes runs before its NMS.  `run()` launches it on the
    current stream; `views()` are the stage's outputs inside the workspace."""

    def __init__(self, geom, cls, reg, iou, img_shapes, scale_factors, rescale):
        cls, reg, iou = list(cls), list(reg), list(iou)
        self.p, self.B, self.dt, self.geom = level_ptrs(geom, cls, reg, iou)
        self.keep = (cls, reg, iou)
        dev = cls[0].device
        L = _lib.lib()
        self.nbytes = L.ia_get_bboxes_workspace_bytes(self.geom.ref(), self.B)
        if self.nbytes == 0:
            raise _lib.IouAwareLibraryError('unsupported geometry / batch for ia_decode_stage')
        self.ws = torch.zeros(int(self.nbytes), dtype=torch.uint8, device=dev)     # WORKSPACE CONTRACT
        self.hw, self.sf = _meta_tensors(img_shapes, scale_factors, dev)
        self.rescale = int(bool(rescale))

    def run(self):
        _lib.check(_lib.lib().ia_decode_stage(self.geom.ref(), C.byref(self.p), self.B, self.dt,
                                              _ptr(self.hw), _ptr(self.sf), self.rescale,
                                              _ptr(self.ws), self.nbytes, _stream()),
                   'ia_decode_stage')

    def views(self):
        off = (C.c_size_t * 8)()
        _lib.check(_lib.lib().ia_get_bboxes_workspace_layout(self.geom.ref(), self.B, C.byref(off)),
                   'workspace_layout')
        g, B = self.geom, self.B

        def view(i, dtype, shape):
            n = int(np.prod(shape))
            return self.ws[off[i]:off[i] + n * 4].view(dtype).view(*shape)
        return dict(rowmax=view(0, torch.float32, (B, g.N)), cand_idx=view(1, torch.int32, (B, g.R)),
                    boxes=view(2, torch.float32, (B, g.R, 4)),
                    scores_t=view(3, torch.float32, (B, g.C, g.Rs)),
                    best_score=view(6, torch.float32, (B, g.R)))


# ----------------------------------------------------------------- stage wrappers
def geometry_for(geom, cls, reg, iou):
    """the geometry object matching the memory order (NCHW / channels-last) of these head
    outputs: pass it to the stage wrappers so that select_topk reads the row maxima in the
    order decode_fuse_rowmax wrote them"""
    return level_ptrs(geom, list(cls), list(reg), list(iou))[3]


def decode_fuse_rowmax(geom, cls, reg, iou, select_ws=None):
    """select_ws: a top-k workspace (select_workspace): the kernel then also leaves the group
    maxima / cleared counters select_topk(..., select_ws) starts from (ia_get_bboxes' chaining)"""
    cls, reg, iou = list(cls), list(reg), list(iou)
    p, B, dt, geom = level_ptrs(geom, cls, reg, iou)
    out = torch.empty((B, geom.N), dtype=torch.float32, device=cls[0].device)
    if select_ws is None:
        _lib.check(_lib.lib().ia_decode_fuse_rowmax(geom.ref(), C.byref(p), B, dt, _ptr(out),
                                                    _stream()), 'ia_decode_fuse_rowmax')
    else:
        _lib.check(_lib.lib().ia_decode_fuse_rowmax_grouped(
            geom.ref(), C.byref(p), B, dt, _ptr(out), _ptr(select_ws), select_ws.numel(),
            _stream()), 'ia_decode_fuse_rowmax_grouped')
    return out


def select_workspace(geom, batch, device):
    nbytes = _lib.lib().ia_select_topk_workspace_bytes(geom.ref(), int(batch))
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


def select_topk(geom, rowmax, select_ws=None):
    """select_ws: the workspace decode_fuse_rowmax(..., select_ws) has just filled"""
    _require_gpu(rowmax, 'rowmax')
    rowmax = rowmax.contiguous()
    B = rowmax.shape[0]
    out = torch.empty((B, geom.R), dtype=torch.int32, device=rowmax.device)
    if select_ws is None:
        ws = select_workspace(geom, B, rowmax.device)
        _lib.check(_lib.lib().ia_select_topk(geom.ref(), _ptr(rowmax), B, _ptr(out), _ptr(ws),
                                             ws.numel(), _stream()), 'ia_select_topk')
    else:
        _lib.check(_lib.lib().ia_select_topk_grouped(geom.ref(), _ptr(rowmax), B, _ptr(out),
                                                     _ptr(select_ws), select_ws.numel(),
                                                     _stream()), 'ia_select_topk_grouped')
    return out


def gather_decode(geom, cls, reg, iou, cand_idx, img_shapes, scale_factors, rescale):
    cls, reg, iou = list(cls), list(reg), list(iou)
    p, B, dt, geom = level_ptrs(geom, cls, reg, iou)
    dev = cls[0].device
    hw, sf = _meta_tensors(img_shapes, scale_factors, dev)
    boxes = torch.empty((B, geom.R, 4), dtype=torch.float32, device=dev)
    scores_t = torch.zeros((B, geom.C, geom.Rs), dtype=torch.float32, device=dev)
    best = torch.empty((B, geom.R), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().ia_gather_decode(geom.ref(), C.byref(p), B, dt, _ptr(cand_idx), _ptr(hw),
                                           _ptr(sf), int(bool(rescale)), _ptr(boxes),
                                           _ptr(scores_t), _ptr(best), _stream()),
               'ia_gather_decode')
    return boxes, scores_t, best


def multiclass_nms(boxes, scores_t, R, score_thr, iou_thr, max_per_img, best_score=None):
    """boxes (B,R,4), scores_t (B,C,Rs) class-major; best_score (B,R) optional."""
    _require_gpu(boxes, 'boxes')
    B, Cn, Rs = scores_t.shape
    dev = boxes.device
    dets = torch.empty((B, max_per_img, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((B, max_per_img), dtype=torch.int32, device=dev)
    rows = torch.empty((B, max_per_img), dtype=torch.int32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    kc = torch.empty((B, Cn), dtype=torch.int32, device=dev)
    kr = torch.empty((B, Cn, Rs), dtype=torch.int32, device=dev)
    nbytes = _lib.lib().ia_multiclass_nms_workspace_bytes(B, int(R), Cn)
    ws = _workspace(dev, nbytes)
    _lib.check(_lib.lib().ia_multiclass_nms(_ptr(boxes.contiguous()), _ptr(scores_t.contiguous()),
                                            _ptr(best_score), B, int(R), Cn, float(score_thr),
                                            float(iou_thr), int(max_per_img), _ptr(ws), nbytes,
                                            _ptr(dets), _ptr(labels), _ptr(rows), _ptr(num),
                                            _ptr(kc), _ptr(kr), _stream()), 'ia_multiclass_nms')
    return dets, labels, rows, num, kc, kr


def multiclass_nms_lazy(boxes, scores_t, R, score_thr, iou_thr, max_per_img, best_score=None,
                        candidates=0):
    """the NMS stage evaluated lazily (csrc/lazynms.hip): -> dets, labels, rows, num"""
    _require_gpu(boxes, 'boxes')
    B, Cn, Rs = scores_t.shape
    dev = boxes.device
    dets = torch.empty((B, max_per_img, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((B, max_per_img), dtype=torch.int32, device=dev)
    rows = torch.empty((B, max_per_img), dtype=torch.int32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    nbytes = _lib.lib().ia_multiclass_nms_lazy_workspace_bytes(B, int(R), Cn)
    ws = _workspace(dev, nbytes)
    _lib.check(_lib.lib().ia_multiclass_nms_lazy(
        _ptr(boxes.contiguous()), _ptr(scores_t.contiguous()), _ptr(best_score), B, int(R), Cn,
        float(score_thr), float(iou_thr), int(max_per_img), int(candidates), _ptr(ws), nbytes,
        _ptr(dets), _ptr(labels), _ptr(rows), _ptr(num), _stream()), 'ia_multiclass_nms_lazy')
    return dets, labels, rows, num


SOFT_METHODS = {'linear': 1, 'gaussian': 2}


def _soft_method(method):
    if method not in SOFT_METHODS:
        raise ValueError('Invalid method for SoftNMS: {}'.format(method))   # nms_wrapper.py:66
    return SOFT_METHODS[method]


def multiclass_soft_nms(boxes, scores_t, R, score_thr, iou_thr, max_per_img, method='linear',
                        sigma=0.5, min_score=1e-3):
    """like multiclass_nms with the soft-NMS operator per class: dets carry decayed scores and
    keep_rows lists each class's survivors in selection order."""
    _require_gpu(boxes, 'boxes')
    B, Cn, Rs = scores_t.shape
    dev = boxes.device
    dets = torch.empty((B, max_per_img, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((B, max_per_img), dtype=torch.int32, device=dev)
    rows = torch.empty((B, max_per_img), dtype=torch.int32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    kc = torch.empty((B, Cn), dtype=torch.int32, device=dev)
    kr = torch.empty((B, Cn, Rs), dtype=torch.int32, device=dev)
    nbytes = _lib.lib().ia_multiclass_soft_nms_workspace_bytes(B, int(R), Cn)
    ws = _workspace(dev, nbytes)
    _lib.check(_lib.lib().ia_multiclass_soft_nms(
        _ptr(boxes.contiguous()), _ptr(scores_t.contiguous()), B, int(R), Cn, float(score_thr),
        float(iou_thr), _soft_method(method), float(sigma), float(min_score), int(max_per_img),
        _ptr(ws), nbytes, _ptr(dets), _ptr(labels), _ptr(rows), _ptr(num), _ptr(kc), _ptr(kr),
        _stream()), 'ia_multiclass_soft_nms')
    return dets, labels, rows, num, kc, kr


def soft_nms_dets(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    """Device soft-NMS on (n,5) fp32 dets -> (new_dets (m,5), inds (m,) int64), selection order."""
    _require_gpu(dets, 'dets')
    n = dets.shape[0]
    code = _soft_method(method)
    if n == 0:
        return dets.new_zeros((0, 5), dtype=torch.float32), dets.new_zeros(0, dtype=torch.long)
    if n > _lib.IA_MAX_CANDIDATES:
        raise _lib.IouAwareLibraryError('soft_nms supports at most %d boxes per call, got %d'
                                        % (_lib.IA_MAX_CANDIDATES, n))
    d = dets.detach().to(torch.float32).contiguous()
    out = torch.empty((n, 5), dtype=torch.float32, device=dets.device)
    inds = torch.empty((n,), dtype=torch.int32, device=dets.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=dets.device)
    _lib.check(_lib.lib().ia_soft_nms(_ptr(d), n, float(iou_thr), code, float(sigma),
                                      float(min_score), _ptr(out), _ptr(inds), _ptr(cnt),
                                      _stream()), 'ia_soft_nms')
    m = int(cnt.item())
    return out[:m], inds[:m].to(torch.long)


def nms_indices(dets, iou_thr):
    """Device NMS on (n,5) dets; returns ascending kept indices (int64, device).  float64 dets take
    the fp64 kernels (nms_cpu_kernel<double>, nms_cpu.cpp:63), everything else is computed in fp32
    like the reference's float instantiation."""
    _require_gpu(dets, 'dets')
    n = dets.shape[0]
    if n == 0:
        return dets.new_zeros(0, dtype=torch.long)
    keep = torch.empty((n,), dtype=torch.int32, device=dets.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=dets.device)
    L = _lib.lib()
    if dets.dtype == torch.float64:
        d = dets.detach().contiguous()
        nbytes = L.ia_nms_f64_workspace_bytes(n)
        if nbytes == 0:
            raise _lib.IouAwareLibraryError('nms on float64 boxes handles at most 16384 boxes, got %d' % n)
        ws = _workspace(dets.device, nbytes)
        _lib.check(L.ia_nms_f64(_ptr(d), n, float(iou_thr), _ptr(keep), _ptr(cnt), _ptr(ws), nbytes,
                                _stream()), 'ia_nms_f64')
    else:
        d = dets.detach().to(torch.float32).contiguous()
        nbytes = L.ia_nms_workspace_bytes(n)
        ws = _workspace(dets.device, nbytes)
        _lib.check(L.ia_nms(_ptr(d), n, float(iou_thr), _ptr(keep), _ptr(cnt), _ptr(ws), nbytes,
                            _stream()), 'ia_nms')
    m = int(cnt.item())
    return keep[:m].to(torch.long)


def channel_affine_act_(x, scale=None, shift=None, residual=None, res_scale=None, res_shift=None,
                        relu=False):
    """In place on a contiguous NCHW tensor: x = act(x*scale[c] + shift[c] [+ residual affine])."""
    _require_gpu(x, 'x')
    N, Cn = x.shape[0], x.shape[1]
    if not x.is_contiguous():
        if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
            if residual is not None:
                if residual.shape != x.shape or residual.dtype != x.dtype:
                    raise ValueError('residual must match x')
                residual = residual.contiguous(memory_format=torch.channels_last)
            _lib.check(_lib.lib().ia_channel_affine_act_nhwc(
                _ptr(x), _dtype_code(x), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(res_scale),
                _ptr(res_shift), int(bool(relu)), x.numel() // Cn, Cn, _stream()),
                'ia_channel_affine_act_nhwc')
            return x
        raise ValueError('channel_affine_act_ needs a contiguous NCHW or channels-last tensor')
    hw = x.numel() // (N * Cn)
    if residual is not None:
        if residual.shape != x.shape or residual.dtype != x.dtype:
            raise ValueError('residual must match x')
        residual = residual.contiguous()
    _lib.check(_lib.lib().ia_channel_affine_act(
        _ptr(x), _dtype_code(x), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(res_scale),
        _ptr(res_shift), int(bool(relu)), N, Cn, hw, _stream()), 'ia_channel_affine_act')
    return x


def upsample2x_add_(fine, coarse):
    """fine += nearest-x2(coarse), in place, channels-last fp32 / bf16 (FPN top-down step)"""
    _require_gpu(fine, 'fine')
    B, Cn, H, W = fine.shape
    vec = 4 if fine.dtype == torch.float32 else 8
    if fine.dtype not in (torch.float32, torch.bfloat16) or coarse.dtype != fine.dtype or Cn % vec \
            or not fine.is_contiguous(memory_format=torch.channels_last) \
            or not coarse.is_contiguous(memory_format=torch.channels_last):
        raise TypeError('upsample2x_add_ needs channels-last fp32 / bf16 tensors of one dtype, '
                        'C % 4 (fp32) / C % 8 (bf16) == 0')
    _lib.check(_lib.lib().ia_upsample2x_add_nhwc_dt(_ptr(fine), _ptr(coarse), _dtype_code(fine), B, H, W,
                                                    int(coarse.shape[2]), int(coarse.shape[3]), Cn,
                                                    _stream()), 'ia_upsample2x_add_nhwc_dt')
    return fine


def affine_relu_maxpool(x, scale, shift):
    """relu(x * scale + shift) followed by MaxPool2d(3, 2, 1) on a channels-last fp32 / bf16 tensor
    (scale / shift fp32)"""
    _require_gpu(x, 'x')
    B, Cn, H, W = x.shape
    vec = 4 if x.dtype == torch.float32 else 8
    if x.dtype not in (torch.float32, torch.bfloat16) or Cn % vec \
            or not x.is_contiguous(memory_format=torch.channels_last):
        raise TypeError('affine_relu_maxpool needs a channels-last fp32 / bf16 tensor, '
                        'C % 4 (fp32) / C % 8 (bf16) == 0')
    scale, shift = scale.float().contiguous(), shift.float().contiguous()
    out = torch.empty((B, Cn, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype,
                      device=x.device, memory_format=torch.channels_last)
    _lib.check(_lib.lib().ia_affine_relu_maxpool_nhwc_dt(_ptr(x), _dtype_code(x), _ptr(scale), _ptr(shift),
                                                         B, H, W, Cn, _ptr(out), _stream()),
               'ia_affine_relu_maxpool_nhwc_dt')
    return out


def conv3x3_bf16_pack(weight, groups=1):
    """(groups * Cout, Cin, 3, 3) weight -> the packed bf16 layout of `conv3x3_bf16_levels`
    (once per model); Cin % 32 == 0, Cout even"""
    _require_gpu(weight, 'weight')
    co, ci, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or ci % 32 or co % groups or (co // groups) % 2 or not 1 <= groups <= 2:
        raise ValueError('conv3x3_bf16 needs a 3x3 kernel, Cin % 32 == 0, an even Cout per group, groups <= 2')
    w = weight.detach().to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()      # (Cout, 3, 3, Cin)
    nbytes = _lib.lib().ia_conv3x3_bf16_packed_bytes(ci, co // groups, groups)
    wp = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=weight.device)
    _lib.check(_lib.lib().ia_conv3x3_bf16_pack(_ptr(w), ci, co // groups, groups, _ptr(wp), _stream()),
               'ia_conv3x3_bf16_pack')
    return wp


def conv3x3_bf16_levels(xs, wp, bias, cout, ys, relu=False, cin=None):
    """3x3 / stride 1 / pad 1 convolution + bias (+ReLU) of bf16 channels-last tensors on the MFMA
    implicit-GEMM kernel (csrc/conv3x3_bf16.hip), all levels and groups in ONE launch.
    xs / ys: list over groups of lists over levels of (B, C, H, W) channels-last bf16 tensors;
    a group's tensors may be channel slices [c0, c0 + n) of wider tensors (the cls / reg halves of
    one activation).  cout / cin: channels per group; wp from conv3x3_bf16_pack; bias fp32
    (groups * cout) or None.  Writes ys in place and returns them."""
    groups, L = len(xs), len(xs[0])
    x0, y0 = xs[0][0], ys[0][0]
    _require_gpu(x0, 'x')
    cin = int(cin if cin is not None else x0.shape[1])
    d = _lib.Conv3x3Desc()
    d.num_levels, d.batch, d.groups = L, int(x0.shape[0]), groups
    d.cin, d.cout = cin, int(cout)

    def pix_stride(t):
        # channels-last (possibly a channel slice): stride of W = channels of the underlying tensor
        if t.dtype != torch.bfloat16 or t.dim() != 4 or t.stride(1) != 1:
            raise TypeError('conv3x3_bf16_levels needs channels-last bf16 tensors')
        # (a 1 x 1 map of a batch: the pixel stride is the image stride -- P6 / P7 of a small input,
        # where the cls / reg halves are channel slices of a 2F-wide activation)
        st = t.stride(3) if t.shape[3] > 1 else (t.stride(2) if t.shape[2] > 1 else
                                                 (t.stride(0) if t.shape[0] > 1 else t.shape[1]))
        if (t.shape[2] > 1 and t.shape[3] > 1 and t.stride(2) != t.shape[3] * st) or \
                (t.shape[0] > 1 and t.stride(0) != t.shape[2] * t.shape[3] * st):
            raise TypeError('conv3x3_bf16_levels: tensor is not a dense channels-last (slice)')
        return int(st)
    d.x_stride, d.y_stride = pix_stride(x0), pix_stride(y0)
    for l in range(L):
        d.H[l], d.W[l] = int(xs[0][l].shape[2]), int(xs[0][l].shape[3])
        for g in range(groups):
            x, y = xs[g][l], ys[g][l]
            if x.shape[1] != cin or y.shape[1] != cout or x.shape[2:] != y.shape[2:] or x.shape[0] != d.batch \
                    or tuple(x.shape[2:]) != (d.H[l], d.W[l]) or pix_stride(x) != d.x_stride \
                    or pix_stride(y) != d.y_stride:
                raise ValueError('conv3x3_bf16_levels: inconsistent tensors at group %d level %d' % (g, l))
            d.x[g][l], d.y[g][l] = x.data_ptr(), y.data_ptr()
    _lib.check(_lib.lib().ia_conv3x3_bf16_levels(C.byref(d), _ptr(wp), _ptr(bias), int(bool(relu)), _stream()),
               'ia_conv3x3_bf16_levels')
    return ys


def conv3x3_bf16(x, wp, bias, cout, relu=False):
    """one tensor, one group: see conv3x3_bf16_levels"""
    y = torch.empty((x.shape[0], cout, x.shape[2], x.shape[3]), dtype=torch.bfloat16, device=x.device,
                    memory_format=torch.channels_last)
    conv3x3_bf16_levels([[x]], wp, bias, cout, [[y]], relu=relu)
    return y


_LT_WS_BYTES = 64 << 20


def linear_bias_act(x, w_kn, bias=None, residual=None, relu=False, w_nk=False):
    """1x1 convolution of a channels-last (B, k, H, W) fp32 tensor as one hipBLASLt GEMM:
    relu?(x . w_kn + bias + residual) -> channels-last (B, n, H, W).  w_kn: (k, n) row-major;
    w_nk=True: the weight is given as (n, k) row-major instead (a (Cout, Cin) convolution weight
    as it is; fp32 only -- the training route)."""
    _require_gpu(x, 'x')
    B, k, H, W = x.shape
    n = int(w_kn.shape[0 if w_nk else 1])
    if x.dtype not in (torch.float32, torch.bfloat16) or \
            not x.is_contiguous(memory_format=torch.channels_last):
        raise TypeError('linear_bias_act needs a channels-last fp32 / bf16 activation')
    if tuple(w_kn.shape) != ((n, k) if w_nk else (k, n)) or not w_kn.is_contiguous() \
            or w_kn.dtype != x.dtype:
        raise ValueError('weight must be a contiguous (k, n) [w_nk: (n, k)] matrix of the '
                         'activation dtype')
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError('bias must be fp32')
    out = torch.empty((B, n, H, W), dtype=x.dtype, device=x.device,
                      memory_format=torch.channels_last)
    if residual is not None and (tuple(residual.shape) != tuple(out.shape) or
                                 not residual.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError('residual must be a channels-last tensor of the output shape')
    if residual is not None and residual.dtype != x.dtype:
        raise TypeError('residual dtype mismatch')
    if x.dtype == torch.float32 and not w_nk and WIDE_1X1 and (k, n) == (128, 512) and residual is not None \
            and B * H * W >= 65536:
        # ResNet stage 2 tail: HBM-bound (619 MB for 17.6 GFLOP at batch 8) -> the streaming kernel's
        # wide-output form, column blocks of 256 channels (csrc/conv1x1_stream.hip, k_conv1x1_wide)
        _lib.check(_lib.lib().ia_conv1x1_wide(_ptr(x), _ptr(w_kn), _ptr(bias), _ptr(residual), _ptr(out),
                                              B * H * W, k, n, int(bool(relu)), _stream()), 'ia_conv1x1_wide')
        return out
    if x.dtype == torch.float32 and not w_nk and STREAM_1X1 and (k, n) in _STREAM_SHAPES \
            and B * H * W >= 65536:
        # ResNet stage 1: HBM-bound products with 16 K weights -> the streaming kernel with the
        # weights in LDS (csrc/conv1x1_stream.hip; 4.3-4.8 TB/s against the library GEMM's 3.3-3.8)
        _lib.check(_lib.lib().ia_conv1x1_stream(_ptr(x), _ptr(w_kn), _ptr(bias), _ptr(residual), _ptr(out),
                                                B * H * W, k, n, int(bool(relu)), _stream()),
                   'ia_conv1x1_stream')
        return out
    _ensure_gemm_table()
    ws = _workspace(x.device, _LT_WS_BYTES)
    if x.dtype == torch.bfloat16:
        if w_nk:
            raise TypeError('w_nk is fp32 only')
        _lib.check(_lib.lib().ia_linear_bias_act_bf16(_ptr(x), _ptr(w_kn), _ptr(bias), _ptr(residual),
                                                      _ptr(out), B * H * W, k, n, int(bool(relu)),
                                                      _ptr(ws), _LT_WS_BYTES, _stream()),
                   'ia_linear_bias_act_bf16')
        return out
    fn = _lib.lib().ia_linear_bias_act_wt if w_nk else _lib.lib().ia_linear_bias_act
    _lib.check(fn(_ptr(x), _ptr(w_kn), _ptr(bias), _ptr(residual), _ptr(out), B * H * W, k, n,
                  int(bool(relu)), _ptr(ws), _LT_WS_BYTES, _stream()), 'ia_linear_bias_act')
    return out


def stem_weight(weight):
    """(64, 3, 7, 7) stem weight -> the (148, 64) fp32 matrix of ia_stem_conv7x7s2
    (row ky * 21 + kx * 3 + c, one zero row of K padding)"""
    if tuple(weight.shape) != (64, 3, 7, 7):
        raise ValueError('the stem kernel covers a (64, 3, 7, 7) weight')
    w = weight.detach().float().permute(2, 3, 1, 0).reshape(147, 64)
    return torch.cat([w, torch.zeros(1, 64, dtype=torch.float32, device=w.device)], 0).contiguous()


def stem_conv(x, w_packed):
    """7x7 / stride 2 / pad 3 convolution 3 -> 64 of a channels-last fp32 image batch on the fp32
    MFMA kernel of csrc/stem.hip -> the raw convolution, channels-last (B, 64, Ho, Wo)"""
    _require_gpu(x, 'x')
    B, C, H, W = x.shape
    if C != 3 or x.dtype != torch.float32 or not x.is_contiguous(memory_format=torch.channels_last) \
            or tuple(w_packed.shape) != (148, 64) or w_packed.dtype != torch.float32 or not w_packed.is_contiguous():
        raise TypeError('stem_conv: channels-last fp32 (B, 3, H, W) input, packed (148, 64) weight')
    y = torch.empty((B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device,
                    memory_format=torch.channels_last)
    _lib.check(_lib.lib().ia_stem_conv7x7s2(_ptr(x), _ptr(w_packed), _ptr(y), B, H, W, _stream()),
               'ia_stem_conv7x7s2')
    return y


def stem_weight_bf16(weight):
    """(64, 3, 7, 7) stem weight -> the fragment-order bf16 packing of ia_stem_conv7x7s2_bf16:
    [step 11][block 4][lane 64][4], k = ky * 24 + 1 + kx * 3 + c, zeros elsewhere"""
    if tuple(weight.shape) != (64, 3, 7, 7):
        raise ValueError('the stem kernel covers a (64, 3, 7, 7) weight')
    w = weight.detach().float()
    wk = torch.zeros((64, 8, 24), dtype=torch.float32, device=w.device)          # (n, ky padded to 8, position in the row of 24)
    wk[:, :7, 1:22] = w.permute(0, 2, 3, 1).reshape(64, 7, 21)                    # k = ky * 24 + 1 + kx * 3 + c (csrc/stem.hip)
    wk = wk.reshape(64, 8 * 24)[:, :176]                                          # 11 steps of 16
    # [n = nb * 16 + m][k = 16 s + 4 q + e] -> [s][nb][lane = q * 16 + m][e]
    wk = wk.reshape(4, 16, 11, 4, 4).permute(2, 0, 3, 1, 4).contiguous()        # (s, nb, q, m, e)
    return wk.reshape(11, 4, 64, 4).to(torch.bfloat16).contiguous()


def stem_conv_bf16(x, w_packed):
    """the stem convolution of a channels-last bf16 image batch (csrc/stem.hip, bf16 MFMA) -> the raw
    convolution, channels-last bf16 (B, 64, Ho, Wo)"""
    _require_gpu(x, 'x')
    B, C, H, W = x.shape
    if C != 3 or x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last) \
            or tuple(w_packed.shape) != (11, 4, 64, 4) or w_packed.dtype != torch.bfloat16 or not w_packed.is_contiguous():
        raise TypeError('stem_conv_bf16: channels-last bf16 (B, 3, H, W) input, packed (11, 4, 64, 4) weight')
    y = torch.empty((B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.bfloat16, device=x.device,
                    memory_format=torch.channels_last)
    _lib.check(_lib.lib().ia_stem_conv7x7s2_bf16(_ptr(x), _ptr(w_packed), _ptr(y), B, H, W, _stream()),
               'ia_stem_conv7x7s2_bf16')
    return y


def conv1x1_chain(x, w_kn, bias, residual, w2_kn, bias2):
    """the boundary between two stage-1 bottlenecks in one pass (csrc/conv1x1_stream.hip,
    k_conv1x1_chain): y = relu(x . w_kn + bias + residual), h = relu(y . w2_kn + bias2) computed from
    the accumulators of the first product -> (y, h), both channels-last fp32"""
    _require_gpu(x, 'x')
    B, k, H, W = x.shape
    n, n2 = int(w_kn.shape[1]), int(w2_kn.shape[1])
    if x.dtype != torch.float32 or not x.is_contiguous(memory_format=torch.channels_last) \
            or (k, n, n2) != _CHAIN_SHAPE or tuple(w_kn.shape) != (k, n) or tuple(w2_kn.shape) != (n, n2) \
            or w_kn.dtype != torch.float32 or w2_kn.dtype != torch.float32 \
            or not w_kn.is_contiguous() or not w2_kn.is_contiguous():
        raise TypeError('conv1x1_chain: fp32 channels-last input, (k, n, n2) = %s' % (_CHAIN_SHAPE,))
    for b_, n_ in ((bias, n), (bias2, n2)):
        if b_ is not None and (b_.dtype != torch.float32 or b_.numel() != n_ or not b_.is_contiguous()):
            raise TypeError('conv1x1_chain: biases are contiguous fp32 vectors of the output widths')
    y = torch.empty((B, n, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    h = torch.empty((B, n2, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if residual is not None and (tuple(residual.shape) != tuple(y.shape) or residual.dtype != torch.float32
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError('residual must be a channels-last fp32 tensor of the output shape')
    _lib.check(_lib.lib().ia_conv1x1_chain(_ptr(x), _ptr(w_kn), _ptr(bias), _ptr(residual), _ptr(w2_kn),
                                           _ptr(bias2), _ptr(y), _ptr(h), B * H * W, k, n, n2, _stream()),
               'ia_conv1x1_chain')
    return y, h


_CHAIN_SHAPE = (64, 256, 64)
CHAIN_1X1 = True                       # fuse._layer_forward chains stage-1 block boundaries


def chain_usable(x, k, n, n2):
    return CHAIN_1X1 and STREAM_1X1 and x.dtype == torch.float32 and (k, n, n2) == _CHAIN_SHAPE \
        and x.shape[0] * x.shape[2] * x.shape[3] >= 65536


_col_cache = {}


def _col_buffer(device, nbytes):
    """the im2col matrix of conv3x3_im2col: one growing buffer per (device, stream)"""
    key = (device.index, stream_id())
    buf = _col_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _col_cache.pop(key, None)
        buf = _col_cache[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return buf


def conv3x3_im2col(x, w_kn, bias=None, stride=2, relu=False):
    """3x3 / pad 1 / stride `stride` convolution of a channels-last (B, C, H, W) fp32 / bf16 tensor
    as im2col (csrc/im2col.hip) + one library GEMM with bias / ReLU in the epilogue: a contraction
    with a fixed reduction order (the library convolution's split-K kernels for these shapes add
    with atomics and give other bits in every run).  w_kn: (9 * C, Cout) row-major, row index
    (dy * 3 + dx) * C + c -- `conv3x3_weight_kn(weight)`."""
    _require_gpu(x, 'x')
    B, Cc, H, W = x.shape
    n = int(w_kn.shape[1])
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_contiguous(memory_format=torch.channels_last):
        raise TypeError('conv3x3_im2col needs a channels-last fp32 / bf16 activation')
    if tuple(w_kn.shape) != (9 * Cc, n) or not w_kn.is_contiguous() or w_kn.dtype != x.dtype:
        raise ValueError('weight must be a contiguous (9 * C, Cout) matrix of the activation dtype')
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError('bias must be fp32')
    dt = _dtype_code(x)
    L = _lib.lib()
    nbytes = L.ia_im2col3x3_bytes(B, H, W, Cc, int(stride), dt)
    if nbytes == 0 or (Cc * x.element_size()) % 16:
        raise ValueError('conv3x3_im2col: C * sizeof(dtype) must be a multiple of 16')
    col = _col_buffer(x.device, nbytes)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    _lib.check(L.ia_im2col3x3_nhwc(_ptr(x), _ptr(col), B, H, W, Cc, int(stride), dt, _stream()),
               'ia_im2col3x3_nhwc')
    out = torch.empty((B, n, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _ensure_gemm_table()
    ws = _workspace(x.device, _LT_WS_BYTES)
    fn = L.ia_linear_bias_act if x.dtype == torch.float32 else L.ia_linear_bias_act_bf16
    _lib.check(fn(_ptr(col), _ptr(w_kn), _ptr(bias), None, _ptr(out), B * Ho * Wo, 9 * Cc, n,
                  int(bool(relu)), _ptr(ws), _LT_WS_BYTES, _stream()), 'ia_linear_bias_act (im2col)')
    return out


def conv3x3_weight_kn(weight, scale=None):
    """(Cout, Cin, 3, 3) convolution weight (x per-output-channel scale) -> the (9 * Cin, Cout) GEMM
    operand of conv3x3_im2col: row (dy * 3 + dx) * Cin + c"""
    w = weight.detach().float()
    if scale is not None:
        w = w * scale.view(-1, 1, 1, 1)
    return w.permute(2, 3, 1, 0).reshape(9 * w.shape[1], w.shape[0]).contiguous()


def conv1x1_strided(x, w_kn, bias=None, residual=None, stride=2, relu=False):
    """1x1 / stride-s convolution of a channels-last (B, k, H, W) tensor -> (B, n, Ho, Wo):
    relu?(x[:, :, ::s, ::s] . w_kn + bias + residual) as one strided-batched library GEMM reading
    the input in place (csrc/gemm.hip, ia_conv1x1_strided)."""
    _require_gpu(x, 'x')
    B, k, H, W = x.shape
    n = int(w_kn.shape[1])
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_contiguous(memory_format=torch.channels_last):
        raise TypeError('conv1x1_strided needs a channels-last fp32 / bf16 activation')
    if tuple(w_kn.shape) != (k, n) or not w_kn.is_contiguous() or w_kn.dtype != x.dtype:
        raise ValueError('weight must be a contiguous (k, n) matrix of the activation dtype')
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError('bias must be fp32')
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((B, n, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if residual is not None and (tuple(residual.shape) != tuple(out.shape) or residual.dtype != x.dtype
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError('residual must be a channels-last tensor of the output shape and dtype')
    _ensure_gemm_table()
    ws = _workspace(x.device, _LT_WS_BYTES)
    _lib.check(_lib.lib().ia_conv1x1_strided(_ptr(x), _ptr(w_kn), _ptr(bias), _ptr(residual), _ptr(out),
                                             B, H, W, k, n, int(stride), int(bool(relu)), _dtype_code(x),
                                             _ptr(ws), _LT_WS_BYTES, _stream()), 'ia_conv1x1_strided')
    return out


_STREAM_SHAPES = ((64, 256), (256, 64), (64, 64))
STREAM_1X1 = True                      # linear_bias_act routes these shapes to conv1x1_stream
WIDE_1X1 = True                        # ... and (128, 512) + residual to ia_conv1x1_wide


def conv1x1_stream(x, w_kn, bias=None, residual=None, relu=False):
    """the stage-1 1x1 convolutions on the streaming MFMA kernel (weights in LDS): fp32
    channels-last, (k, n) in _STREAM_SHAPES; same contract as linear_bias_act"""
    _require_gpu(x, 'x')
    B, k, H, W = x.shape
    n = int(w_kn.shape[1])
    if x.dtype != torch.float32 or not x.is_contiguous(memory_format=torch.channels_last) \
            or (k, n) not in _STREAM_SHAPES or tuple(w_kn.shape) != (k, n) or not w_kn.is_contiguous():
        raise TypeError('conv1x1_stream: fp32 channels-last input, (k, n) in %s' % (_STREAM_SHAPES,))
    out = torch.empty((B, n, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if residual is not None and (tuple(residual.shape) != tuple(out.shape) or residual.dtype != torch.float32
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError('residual must be a channels-last fp32 tensor of the output shape')
    _lib.check(_lib.lib().ia_conv1x1_stream(_ptr(x), _ptr(w_kn), _ptr(bias), _ptr(residual), _ptr(out),
                                            B * H * W, k, n, int(bool(relu)), _stream()), 'ia_conv1x1_stream')
    return out


def gemm_tuning(mode=None):
    """How the library kernel of a new GEMM shape is chosen (csrc/gemm.hip):
      'frozen' (default)  the committed tuning table, else the library heuristic's first result --
                          nothing is timed, the same bits in every run;
      'heuristic' / 'all' OFFLINE tuning: the first call of a shape times the heuristic's top 16 /
                          every library kernel that supports it (~0.3 s per shape).
    Returns the mode in force before the call; None only queries."""
    _ensure_gemm_table()
    code = {'heuristic': 0, 'all': 1, 'frozen': 2, None: -1}[mode]
    return ('heuristic', 'all', 'frozen')[_lib.lib().ia_gemm_tuning(code)]


GEMM_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuning', 'hipblaslt_gfx950.json')
_gemm_table_state = None


def gemm_table_load(path=None, strict_version=True):
    """Install a tuning table written by `gemm_table_save` (tools/tune_gemm.py).  Entries measured
    with another hipBLASLt version are ignored (solution indices belong to a library build).
    -> dict(path, entries, library_version, table_version, used)"""
    import json
    global _gemm_table_state
    path = path or GEMM_TABLE
    L = _lib.lib()
    L.ia_gemm_table_clear()
    state = dict(path=path, entries=0, library_version=L.ia_gemm_library_version(),
                 table_version=None, used=False)
    if os.path.exists(path):
        with open(path) as fh:
            tab = json.load(fh)
        state['table_version'] = tab.get('hipblaslt_version')
        if not strict_version or state['table_version'] == state['library_version']:
            for e in tab['entries']:
                _lib.check(L.ia_gemm_table_add(*[int(v) for v in e[:7]]), 'ia_gemm_table_add')
            state['entries'], state['used'] = len(tab['entries']), True
    _gemm_table_state = state
    return state


def _ensure_gemm_table():
    if _gemm_table_state is None:
        gemm_table_load()


def gemm_table_dump():
    """[(m, n, k, flags, batch, dtype, solution index)] of every GEMM shape resolved so far"""
    L = _lib.lib()
    n = L.ia_gemm_table_dump(None, 0)
    buf = np.zeros((max(n, 1), 7), np.int64)
    n = L.ia_gemm_table_dump(buf.ctypes.data, int(buf.shape[0]))
    return [tuple(int(v) for v in r) for r in buf[:n]]


def gemm_table_stats():
    """dict(hits=, misses=, stale=): shapes served by the table / by the heuristic's first result /
    table entries the library no longer supports"""
    buf = np.zeros(3, np.int64)
    _lib.check(_lib.lib().ia_gemm_table_stats(buf.ctypes.data), 'ia_gemm_table_stats')
    return dict(hits=int(buf[0]), misses=int(buf[1]), stale=int(buf[2]),
                **{k: v for k, v in (_gemm_table_state or {}).items() if k != 'path'})


def gemm_table_save(path=None, merge=True):
    """write the solutions in use to a table file (offline tuning); merge=True keeps the entries
    of an existing file for shapes this process did not run"""
    import json
    path = path or GEMM_TABLE
    ver = _lib.lib().ia_gemm_library_version()
    rows = {r[:6]: r[6] for r in gemm_table_dump()}
    if merge and os.path.exists(path):
        with open(path) as fh:
            old = json.load(fh)
        if old.get('hipblaslt_version') == ver:
            for e in old['entries']:
                rows.setdefault(tuple(int(v) for v in e[:6]), int(e[6]))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as fh:
        json.dump(dict(hipblaslt_version=ver, arch='gfx950',
                       key='m, n, k, flags, batch, dtype (column-major terms of csrc/gemm.hip), solution index',
                       entries=[list(k) + [v] for k, v in sorted(rows.items())]), fh, indent=0)
    return len(rows)


def gemm_tn(g, x):
    """g (rows, n) or (batch, rows, n), x (rows, k) or (batch, rows, k), fp32 contiguous ->
    g^T x: (n, k) / (batch, n, k).  The weight-gradient product of the convolution nodes: a
    reduction over `rows` with a small result (library GEMM, split-K candidates timed)."""
    _require_gpu(g, 'g')
    if g.dtype != torch.float32 or x.dtype != torch.float32 or g.dim() != x.dim() \
            or g.dim() not in (2, 3) or g.shape[:-1] != x.shape[:-1] \
            or not g.is_contiguous() or not x.is_contiguous():
        raise ValueError('gemm_tn needs contiguous fp32 (.., rows, n) and (.., rows, k)')
    batch = int(g.shape[0]) if g.dim() == 3 else 1
    rows, n, k = int(g.shape[-2]), int(g.shape[-1]), int(x.shape[-1])
    out = torch.empty(((batch, n, k) if g.dim() == 3 else (n, k)), dtype=torch.float32,
                      device=g.device)
    _ensure_gemm_table()
    ws = _workspace(g.device, _LT_WS_BYTES)
    _lib.check(_lib.lib().ia_gemm_tn(_ptr(g), _ptr(x), _ptr(out), batch, rows, n, k, _ptr(ws),
                                     _LT_WS_BYTES, _stream()), 'ia_gemm_tn')
    return out


def pack_grouped_weight(weight, scale=None):
    """(C, C/groups, 3, 3) grouped conv weight (+ per-output-channel scale) -> the per-lane MFMA
    operand layout of csrc/gconv.hip, on the weight's device.  Host-side arrangement (once per
    fuse), through the C-ABI helper."""
    w = weight.detach().to('cpu', torch.float32).contiguous()
    Cn, cg = int(w.shape[0]), int(w.shape[1])
    groups = Cn // cg
    nb = 2 if cg == 32 else 1
    out = torch.empty(Cn // (16 * nb) * 9 * nb * 4 * nb * 64, dtype=torch.float32)
    sc = None if scale is None else scale.detach().to('cpu', torch.float32).contiguous()
    _lib.check(_lib.lib().ia_grouped_conv3x3_pack(w.data_ptr(), None if sc is None else sc.data_ptr(),
                                                  Cn, groups, out.data_ptr()),
               'ia_grouped_conv3x3_pack')
    return out.to(weight.device)


def grouped_conv3x3(x, wpack, bias, groups, stride=1, relu=False):
    """channels-last fp32 (B, C, H, W) -> (B, C, Ho, Wo), grouped 3x3 / pad 1 conv + bias (+ReLU)"""
    _require_gpu(x, 'x')
    if x.dtype != torch.float32 or not x.is_contiguous(memory_format=torch.channels_last):
        raise ValueError('grouped_conv3x3 needs a channels-last fp32 tensor')
    B, Cn, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, Cn, Ho, Wo), dtype=torch.float32, device=x.device,
                    memory_format=torch.channels_last)
    _lib.check(_lib.lib().ia_grouped_conv3x3_nhwc(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(y), B, H, W,
                                                  Cn, int(groups), int(stride), int(bool(relu)),
                                                  _stream()), 'ia_grouped_conv3x3_nhwc')
    return y


def test_math(op, x, y=None):
    _require_gpu(x, 'x')
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.lib().ia_test_math(int(op), _ptr(x), _ptr(y.contiguous() if y is not None else None),
                                       _ptr(out), x.numel(), _stream()), 'ia_test_math')
    return out


# ----------------------------------------------------------------- training losses
def _gdev(g):
    """autograd's grad_output as a 1-element fp32 device tensor (no host sync)"""
    return g.detach().reshape(-1)[:1].to(torch.float32).contiguous()


class _FocalLossFn(torch.autograd.Function):
    """sum over the level of py_sigmoid_focal_loss on NCHW logits."""

    @staticmethod
    def forward(ctx, cls, labels, label_weights, A, gamma, alpha):
        _require_gpu(cls, 'cls_score')
        cls = cls.contiguous()
        B, ch, H, W = cls.shape
        Cn = ch // A
        labels = labels.contiguous().view(-1).to(torch.int64)
        lw = label_weights.contiguous().view(-1).to(torch.float32)
        acc = torch.zeros(_lib.IA_LOSS_SLOTS, dtype=torch.float64, device=cls.device)
        _lib.check(_lib.lib().ia_focal_loss_fwd(_ptr(cls), _dtype_code(cls), _ptr(labels), _ptr(lw),
                                                B, A, Cn, H * W, float(gamma), float(alpha),
                                                _ptr(acc), _stream()), 'ia_focal_loss_fwd')
        ctx.save_for_backward(cls, labels, lw)
        ctx.cfg = (A, Cn, float(gamma), float(alpha))
        return acc.sum().reshape(1).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        cls, labels, lw = ctx.saved_tensors
        A, Cn, gamma, alpha = ctx.cfg
        B, ch, H, W = cls.shape
        grad = torch.empty(cls.shape, dtype=torch.float32, device=cls.device)
        _lib.check(_lib.lib().ia_focal_loss_bwd(_ptr(cls), _dtype_code(cls), _ptr(labels), _ptr(lw),
                                                B, A, Cn, H * W, gamma, alpha, 1.0, _ptr(_gdev(g)),
                                                _ptr(grad), _stream()), 'ia_focal_loss_bwd')
        return grad.to(cls.dtype), None, None, None, None, None


def focal_loss_sum(cls, labels, label_weights, num_anchors, gamma=2.0, alpha=0.25):
    return _FocalLossFn.apply(cls, labels, label_weights, num_anchors, gamma, alpha)


class _SmoothL1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, A, beta):
        _require_gpu(pred, 'bbox_pred')
        pred = pred.contiguous()
        B, ch, H, W = pred.shape
        target = target.contiguous().to(torch.float32)
        weight = weight.contiguous().to(torch.float32)
        acc = torch.zeros(_lib.IA_LOSS_SLOTS, dtype=torch.float64, device=pred.device)
        _lib.check(_lib.lib().ia_smooth_l1_fwd(_ptr(pred), _dtype_code(pred), _ptr(target),
                                               _ptr(weight), B, A, H * W, float(beta), _ptr(acc),
                                               _stream()), 'ia_smooth_l1_fwd')
        ctx.save_for_backward(pred, target, weight)
        ctx.cfg = (A, float(beta))
        return acc.sum().reshape(1).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        pred, target, weight = ctx.saved_tensors
        A, beta = ctx.cfg
        B, ch, H, W = pred.shape
        grad = torch.empty(pred.shape, dtype=torch.float32, device=pred.device)
        _lib.check(_lib.lib().ia_smooth_l1_bwd(_ptr(pred), _dtype_code(pred), _ptr(target),
                                               _ptr(weight), B, A, H * W, beta, 1.0, _ptr(_gdev(g)),
                                               _ptr(grad), _stream()), 'ia_smooth_l1_bwd')
        return grad.to(pred.dtype), None, None, None, None


def smooth_l1_sum(pred, target, weight, num_anchors, beta):
    return _SmoothL1Fn.apply(pred, target, weight, num_anchors, beta)


class _IouBceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bbox_pred, iou_pred, bbox_targets, bbox_weights, geom, level, attach_target,
                want_iou=False):
        _require_gpu(bbox_pred, 'bbox_pred')
        bbox_pred, iou_pred = bbox_pred.contiguous(), iou_pred.contiguous()
        if bbox_pred.dtype != iou_pred.dtype:
            raise TypeError('bbox_pred / iou_pred dtype mismatch')
        B = bbox_pred.shape[0]
        bt = bbox_targets.contiguous().to(torch.float32)
        bw = bbox_weights.contiguous().to(torch.float32)
        acc = torch.zeros(_lib.IA_LOSS_SLOTS, dtype=torch.float64, device=bbox_pred.device)
        iou = torch.empty(iou_pred.numel(), dtype=torch.float32,
                          device=bbox_pred.device) if want_iou else None
        _lib.check(_lib.lib().ia_iou_bce_fwd(geom.ref(), int(level), _ptr(bbox_pred), _ptr(iou_pred),
                                             _dtype_code(bbox_pred), _ptr(bt), _ptr(bw), B,
                                             _ptr(iou), _ptr(acc), _stream()),
                   'ia_iou_bce_fwd')
        ctx.save_for_backward(bbox_pred, iou_pred, bt, bw)
        ctx.cfg = (geom, int(level), bool(attach_target))
        loss = acc.sum().reshape(1).to(torch.float32)
        if not want_iou:
            return loss
        ctx.mark_non_differentiable(iou)
        return loss, iou

    @staticmethod
    def backward(ctx, g, *unused):
        bbox_pred, iou_pred, bt, bw = ctx.saved_tensors
        geom, level, attach = ctx.cfg
        B = bbox_pred.shape[0]
        g_iou = torch.empty(iou_pred.shape, dtype=torch.float32, device=iou_pred.device)
        g_box = torch.empty(bbox_pred.shape, dtype=torch.float32,
                            device=bbox_pred.device) if attach else None
        _lib.check(_lib.lib().ia_iou_bce_bwd(geom.ref(), level, _ptr(bbox_pred), _ptr(iou_pred),
                                             _dtype_code(bbox_pred), _ptr(bt), _ptr(bw), B,
                                             1.0, _ptr(_gdev(g)), _ptr(g_iou), _ptr(g_box),
                                             _stream()), 'ia_iou_bce_bwd')
        return (g_box.to(bbox_pred.dtype) if attach else None, g_iou.to(iou_pred.dtype), None,
                None, None, None, None, None)


def iou_bce_sum(bbox_pred, iou_pred, bbox_targets, bbox_weights, geom, level, attach_target=True,
                return_iou=False):
    """-> loss sum (1,), or (loss sum, iou targets (B*N_l) detached) with return_iou"""
    return _IouBceFn.apply(bbox_pred, iou_pred, bbox_targets, bbox_weights, geom, level,
                           attach_target, return_iou)


# ----------------------------------------------------------------- IoU-balanced variants
class _FocalBalancedFn(torch.autograd.Function):
    """iou_balanced_sigmoid_focal_loss (reference losses.py:309-374) summed over one level:
    S0 + normalizer * S2, normalizer = S1 / (S2 + 1e-6) -- all on the device."""

    @staticmethod
    def forward(ctx, cls, labels, label_weights, iou, A, gamma, alpha, eta):
        _require_gpu(cls, 'cls_score')
        cls = cls.contiguous()
        B, ch, H, W = cls.shape
        Cn = ch // A
        labels = labels.contiguous().view(-1).to(torch.int64)
        lw = label_weights.contiguous().view(-1).to(torch.float32)
        iou = iou.detach().contiguous().view(-1).to(torch.float32)
        acc = torch.zeros((3, _lib.IA_LOSS_SLOTS), dtype=torch.float64, device=cls.device)
        _lib.check(_lib.lib().ia_focal_loss_balanced_fwd(
            _ptr(cls), _dtype_code(cls), _ptr(labels), _ptr(lw), _ptr(iou), B, A, Cn, H * W,
            float(gamma), float(alpha), float(eta), _ptr(acc), _stream()),
            'ia_focal_loss_balanced_fwd')
        S = acc.sum(1).to(torch.float32)
        normalizer = (S[1] / (S[2] + 1e-6)).reshape(1).contiguous()
        ctx.save_for_backward(cls, labels, lw, iou, normalizer)
        ctx.cfg = (A, Cn, float(gamma), float(alpha), float(eta))
        return (S[0] + normalizer[0] * S[2]).reshape(1)

    @staticmethod
    def backward(ctx, g):
        cls, labels, lw, iou, normalizer = ctx.saved_tensors
        A, Cn, gamma, alpha, eta = ctx.cfg
        B, ch, H, W = cls.shape
        grad = torch.empty(cls.shape, dtype=torch.float32, device=cls.device)
        _lib.check(_lib.lib().ia_focal_loss_balanced_bwd(
            _ptr(cls), _dtype_code(cls), _ptr(labels), _ptr(lw), _ptr(iou), B, A, Cn, H * W, gamma,
            alpha, eta, _ptr(normalizer), 1.0, _ptr(_gdev(g)), _ptr(grad), _stream()),
            'ia_focal_loss_balanced_bwd')
        return grad.to(cls.dtype), None, None, None, None, None, None, None


def focal_loss_balanced_sum(cls, labels, label_weights, iou, num_anchors, gamma=2.0, alpha=0.25,
                            eta=1.5):
    return _FocalBalancedFn.apply(cls, labels, label_weights, iou, num_anchors, gamma, alpha, eta)


class _SmoothL1BalancedFn(torch.autograd.Function):
    """weighted_iou_balanced_smoothl1 (reference losses.py:416-458) summed over one level."""

    @staticmethod
    def forward(ctx, pred, target, weight, iou, A, beta, delta):
        _require_gpu(pred, 'bbox_pred')
        pred = pred.contiguous()
        B, ch, H, W = pred.shape
        target = target.contiguous().to(torch.float32)
        weight = weight.contiguous().to(torch.float32)
        iou = iou.detach().contiguous().view(-1).to(torch.float32)
        acc = torch.zeros(_lib.IA_LOSS_SLOTS, dtype=torch.float64, device=pred.device)
        _lib.check(_lib.lib().ia_smooth_l1_balanced_fwd(
            _ptr(pred), _dtype_code(pred), _ptr(target), _ptr(weight), _ptr(iou), B, A, H * W,
            float(beta), float(delta), _ptr(acc), _stream()), 'ia_smooth_l1_balanced_fwd')
        ctx.save_for_backward(pred, target, weight, iou)
        ctx.cfg = (A, float(beta), float(delta))
        return acc.sum().reshape(1).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        pred, target, weight, iou = ctx.saved_tensors
        A, beta, delta = ctx.cfg
        B, ch, H, W = pred.shape
        grad = torch.empty(pred.shape, dtype=torch.float32, device=pred.device)
        _lib.check(_lib.lib().ia_smooth_l1_balanced_bwd(
            _ptr(pred), _dtype_code(pred), _ptr(target), _ptr(weight), _ptr(iou), B, A, H * W, beta,
            delta, 1.0, _ptr(_gdev(g)), _ptr(grad), _stream()), 'ia_smooth_l1_balanced_bwd')
        return grad.to(pred.dtype), None, None, None, None, None, None


def smooth_l1_balanced_sum(pred, target, weight, iou, num_anchors, beta, delta):
    return _SmoothL1BalancedFn.apply(pred, target, weight, iou, num_anchors, beta, delta)


def iou_targets(bbox_pred, iou_pred, bbox_targets, bbox_weights, geom, level):
    """The IoU regression targets of one level (B*N_l) -- for tests / analysis."""
    B = bbox_pred.shape[0]
    h, w = geom.featmap_sizes[level]
    out = torch.empty(B * h * w * geom.A, dtype=torch.float32, device=bbox_pred.device)
    acc = torch.zeros(_lib.IA_LOSS_SLOTS, dtype=torch.float64, device=bbox_pred.device)
    _lib.check(_lib.lib().ia_iou_bce_fwd(geom.ref(), int(level), _ptr(bbox_pred.contiguous()),
                                         _ptr(iou_pred.contiguous()), _dtype_code(bbox_pred),
                                         _ptr(bbox_targets.contiguous()),
                                         _ptr(bbox_weights.contiguous()), B, _ptr(out), _ptr(acc),
                                         _stream()), 'ia_iou_bce_fwd')
    return out, acc.sum().reshape(1)


def anchor_targets(geom, gt_bboxes, gt_labels, pad_shapes, pos_iou_thr, neg_iou_thr, min_pos_iou,
                   pos_weight):
    """Device target assignment for a batch.  gt_bboxes: list of (G_i,4) device tensors (G_i >= 1),
    gt_labels: list of (G_i,) int64 tensors or None.  Returns per-level lists
    labels[(B,N_l) i64], label_weights[(B,N_l)], bbox_targets[(B,N_l,4)], bbox_weights[(B,N_l,4)]
    and counts (B,2) int32 (positives, negatives) -- all on the device, no host sync."""
    B = len(gt_bboxes)
    dev = gt_bboxes[0].device
    _require_gpu(gt_bboxes[0], 'gt_bboxes')
    sizes = [int(g_.shape[0]) for g_ in gt_bboxes]
    if min(sizes) < 1:
        raise ValueError('No gt or bboxes')
    gmax = max(sizes)
    vhw = []
    for (h, w) in [tuple(p[:2]) for p in pad_shapes]:
        vhw.append([[min(int(np.ceil(h / s)), fh), min(int(np.ceil(w / s)), fw)]
                    for s, (fh, fw) in zip(geom.strides, geom.featmap_sizes)])
    N = geom.N
    labels = torch.empty(B * N, dtype=torch.int64, device=dev)
    lw = torch.empty(B * N, dtype=torch.float32, device=dev)
    bt = torch.empty(B * N * 4, dtype=torch.float32, device=dev)
    bw = torch.empty(B * N * 4, dtype=torch.float32, device=dev)
    counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
    scratch = torch.empty((B, gmax), dtype=torch.int32, device=dev)
    if B <= _lib.IA_MAX_TARGET_BATCH:
        # gt tensors stay where the loader put them: their pointers and sizes ride in the kernel
        # arguments (no padded staging copy, no host-to-device transfer)
        # (.to(dev, ...) is a no-op for tensors already there; a host or other-device tensor is
        # copied over instead of handing its pointer to the kernel)
        keep = [g_.to(dev, torch.float32).contiguous() for g_ in gt_bboxes]
        gp = (C.c_void_p * B)(*[g_.data_ptr() for g_ in keep])
        if gt_labels is not None:
            keep_l = [l_.to(dev, torch.int64).contiguous() for l_ in gt_labels]
            lp = (C.c_void_p * B)(*[l_.data_ptr() for l_ in keep_l])
        else:
            lp = None
        ng = (C.c_int32 * B)(*sizes)
        vh = (C.c_int32 * (B * geom.L * 2))(*[v for img in vhw for lv in img for v in lv])
        _lib.check(_lib.lib().ia_anchor_targets_ptrs(
            geom.ref(), gp, lp, ng, B, vh, float(pos_iou_thr), float(neg_iou_thr),
            float(min_pos_iou), float(pos_weight), _ptr(scratch), _ptr(labels), _ptr(lw), _ptr(bt),
            _ptr(bw), _ptr(counts), _stream()), 'ia_anchor_targets_ptrs')
    else:
        boxes = torch.zeros((B, gmax, 4), dtype=torch.float32, device=dev)
        labs = torch.zeros((B, gmax), dtype=torch.int64, device=dev) \
            if gt_labels is not None else None
        for i, g_ in enumerate(gt_bboxes):
            boxes[i, :sizes[i]] = g_.to(torch.float32)
            if labs is not None:
                labs[i, :sizes[i]] = gt_labels[i].to(torch.int64)
        num_gt = torch.tensor(sizes, dtype=torch.int32).to(dev, non_blocking=True)
        vhw_t = torch.tensor(vhw, dtype=torch.int32).to(dev, non_blocking=True)
        _lib.check(_lib.lib().ia_anchor_targets(
            geom.ref(), _ptr(boxes), _ptr(labs), _ptr(num_gt), B, gmax, _ptr(vhw_t),
            float(pos_iou_thr), float(neg_iou_thr), float(min_pos_iou), float(pos_weight),
            _ptr(scratch), _ptr(labels), _ptr(lw), _ptr(bt), _ptr(bw), _ptr(counts), _stream()),
            'ia_anchor_targets')
    out = ([], [], [], [])
    off = 0
    for n_l in geom.level_anchors:
        sl = slice(B * off, B * (off + n_l))
        out[0].append(labels[sl].view(B, n_l))
        out[1].append(lw[sl].view(B, n_l))
        out[2].append(bt[4 * B * off:4 * B * (off + n_l)].view(B, n_l, 4))
        out[3].append(bw[4 * B * off:4 * B * (off + n_l)].view(B, n_l, 4))
        off += n_l
    return out[0], out[1], out[2], out[3], counts


class LevelLosses(list):
    """per-level (1,)-shaped loss tensors, like the reference's lists, plus `.total` = their sum
    computed by the finalize kernel (differentiable): train.parse_losses adds three tensors
    instead of reducing fifteen."""
    total = None


_ZERO1 = {}


def _zero1(dev):
    z = _ZERO1.get(dev)
    if z is None:
        z = _ZERO1[dev] = torch.zeros(1, dtype=torch.float32, device=dev)
    return z


class _HeadLossFn(torch.autograd.Function):
    """the three losses of every level: 3 launches forward, 2 backward (csrc/headloss.hip).
    Outputs: 3L per-level (1,) tensors and 3 totals, views of one result vector."""

    @staticmethod
    def forward(ctx, geom, targets, cfg, *outs):
        L = geom.L
        cls, reg, iou = outs[:L], outs[L:2 * L], outs[2 * L:3 * L]
        for t in outs:
            _require_gpu(t, 'head output')
        cls, reg, iou = [[t.contiguous() for t in x] for x in (cls, reg, iou)]
        B, dev, dt = cls[0].shape[0], cls[0].device, _dtype_code(cls[0])
        if any(_dtype_code(t) != dt for t in cls + reg + iou):
            raise TypeError('head outputs must share one dtype')
        p = LevelPtrs()
        for l in range(L):
            p.cls[l], p.reg[l], p.iou[l] = cls[l].data_ptr(), reg[l].data_ptr(), iou[l].data_ptr()
        labels, lw, bt, bw, counts, avg = targets
        labels = [t.contiguous().to(torch.int64) for t in labels]
        lw = [t.contiguous().to(torch.float32) for t in lw]
        bt = [t.contiguous().to(torch.float32) for t in bt]
        bw = [t.contiguous().to(torch.float32) for t in bw]
        ht = _lib.HeadTargets()
        for l in range(L):
            ht.labels[l], ht.label_weights[l] = labels[l].data_ptr(), lw[l].data_ptr()
            ht.bbox_targets[l], ht.bbox_weights[l] = bt[l].data_ptr(), bw[l].data_ptr()
        avg_dev = None
        if counts is not None:
            ht.counts = counts.data_ptr()
        elif torch.is_tensor(avg):
            avg_dev = avg.detach().reshape(-1)[:1].to(torch.float32).contiguous()
            ht.avg_factor_dev = avg_dev.data_ptr()
        else:
            ht.avg_factor = float(avg)
        hc = _lib.HeadLossCfg(*cfg)
        res = torch.empty(3 * L + 4, dtype=torch.float32, device=dev)
        g = geom.with_layout(_lib.IA_LAYOUT_NCHW)
        nbytes = _lib.lib().ia_head_loss_workspace_bytes(g.ref(), B)
        if nbytes == 0:
            raise _lib.IouAwareLibraryError('unsupported geometry / batch for ia_head_loss')
        # own buffer (not the shared inference workspace): it carries the packed targets from
        # the forward to the backward call
        ws = _own_workspace(dev, nbytes)
        _lib.check(_lib.lib().ia_head_loss_fwd(g.ref(), C.byref(p), dt, B, C.byref(ht),
                                               C.byref(hc), _ptr(ws), nbytes, _ptr(res),
                                               _stream()), 'ia_head_loss_fwd')
        ctx.ws = ws
        ctx.geom, ctx.cfg, ctx.B, ctx.dt = g, hc, B, dt
        ctx.keep = (cls, reg, iou, labels, lw, bt, bw, counts, avg_dev, p, ht)
        ctx.res = res
        ctx.set_materialize_grads(False)
        return tuple(res[:3 * L + 3].view(3 * L + 3, 1).unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        cls, reg, iou, labels, lw, bt, bw, counts, avg_dev, p, ht = ctx.keep
        L, dev = ctx.geom.L, cls[0].device
        z = _zero1(dev)
        gin = torch.cat([z if g is None else g.detach().reshape(1).to(torch.float32) for g in gs])
        gp = LevelPtrs()
        grads = [[torch.empty(t.shape, dtype=torch.float32, device=dev) for t in x]
                 for x in (cls, reg, iou)]
        for l in range(L):
            gp.cls[l], gp.reg[l], gp.iou[l] = (grads[0][l].data_ptr(), grads[1][l].data_ptr(),
                                               grads[2][l].data_ptr())
        _lib.check(_lib.lib().ia_head_loss_bwd(ctx.geom.ref(), C.byref(p), ctx.dt, ctx.B,
                                               C.byref(ht), C.byref(ctx.cfg), _ptr(ctx.ws),
                                               _ptr(ctx.res), _ptr(gin), C.byref(gp), _stream()),
                   'ia_head_loss_bwd')
        out = [g if g.dtype == t.dtype else g.to(t.dtype)
               for x, gx in zip((cls, reg, iou), grads) for t, g in zip(x, gx)]
        return (None, None, None) + tuple(out)


def _pix_stride(t):
    """pixel stride (elements) of a channels-last-like (B, C, H, W) tensor -- channels-last itself,
    or a channel slice of a channels-last tensor -- else None"""
    if t.dim() != 4 or not t.is_cuda or t.dtype != torch.float32:
        return None
    B, Cn, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if Cn > 1 and sc != 1:
        return None
    ps = sw if W > 1 else (sh if H > 1 else (sb if B > 1 else Cn))
    if ps < Cn or (W > 1 and sw != ps) or (H > 1 and sh != W * ps) or (B > 1 and sb != H * W * ps):
        return None
    return int(ps)


def _shared_base(r, i):
    """reg / iou as channel slices of ONE channels-last tensor (the training head's 48-channel
    output): -> that tensor, else None"""
    b = r._base
    if b is None or b is not i._base or b.dim() != 4 or b.dtype != torch.float32 \
            or not b.is_contiguous(memory_format=torch.channels_last) \
            or tuple(b.shape[2:]) != tuple(r.shape[2:]) or b.shape[0] != r.shape[0]:
        return None
    ps = b.shape[1]
    if _pix_stride(r) != ps or _pix_stride(i) != ps:
        return None
    # the backward writes the gradient of reg | iou into one tensor shaped like `b` and zeroes
    # only what lies BEHIND them: the slices must be channels [0, n_reg) and [n_reg, n_reg + n_iou)
    n_reg = r.shape[1]
    if r.storage_offset() != b.storage_offset() or \
            i.storage_offset() != b.storage_offset() + n_reg or n_reg + i.shape[1] > ps:
        return None
    return b


class _HeadLossNhwcFn(torch.autograd.Function):
    """the three losses of every level on channels-last head outputs (csrc/headloss.hip,
    k_focal_nhwc / k_box_nhwc): no layout copies, no packed targets.  Inputs: cls[L], then either
    reg[L] + iou[L], or (fused) the L wider tensors reg / iou are channel slices of -- their
    gradient is then written in place into one tensor of that shape."""

    @staticmethod
    def forward(ctx, geom, targets, cfg, views, *outs):
        L = geom.L
        cls = list(outs[:L])
        fused = views is not None
        if fused:
            bases = list(outs[L:2 * L])
            reg, iou = views
        else:
            bases = None
            reg, iou = list(outs[L:2 * L]), list(outs[2 * L:3 * L])
        B, dev = cls[0].shape[0], cls[0].device
        p, st = LevelPtrs(), _lib.LevelPixStrides()
        for l in range(L):
            p.cls[l], p.reg[l], p.iou[l] = cls[l].data_ptr(), reg[l].data_ptr(), iou[l].data_ptr()
            st.cls[l], st.reg[l], st.iou[l] = _pix_stride(cls[l]), _pix_stride(reg[l]), _pix_stride(iou[l])
        labels, lw, bt, bw, counts, avg = targets
        labels = [t.contiguous().to(torch.int64) for t in labels]
        lw = [t.contiguous().to(torch.float32) for t in lw]
        bt = [t.contiguous().to(torch.float32) for t in bt]
        bw = [t.contiguous().to(torch.float32) for t in bw]
        ht = _lib.HeadTargets()
        for l in range(L):
            ht.labels[l], ht.label_weights[l] = labels[l].data_ptr(), lw[l].data_ptr()
            ht.bbox_targets[l], ht.bbox_weights[l] = bt[l].data_ptr(), bw[l].data_ptr()
        avg_dev = None
        if counts is not None:
            ht.counts = counts.data_ptr()
        elif torch.is_tensor(avg):
            avg_dev = avg.detach().reshape(-1)[:1].to(torch.float32).contiguous()
            ht.avg_factor_dev = avg_dev.data_ptr()
        else:
            ht.avg_factor = float(avg)
        hc = _lib.HeadLossCfg(*cfg)
        res = torch.empty(3 * L + 4, dtype=torch.float32, device=dev)
        nbytes = 8 * 3 * L * _lib.IA_LOSS_SLOTS
        ws = _workspace(dev, nbytes)
        _lib.check(_lib.lib().ia_head_loss_fwd_nhwc(geom.ref(), C.byref(p), C.byref(st), B,
                                                    C.byref(ht), C.byref(hc), _ptr(ws), nbytes,
                                                    _ptr(res), _stream()), 'ia_head_loss_fwd_nhwc')
        ctx.geom, ctx.cfg, ctx.B, ctx.fused = geom, hc, B, fused
        ctx.keep = (cls, reg, iou, bases, labels, lw, bt, bw, counts, avg_dev, p, st, ht)
        ctx.res = res
        ctx.set_materialize_grads(False)
        return tuple(res[:3 * L + 3].view(3 * L + 3, 1).unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        cls, reg, iou, bases, labels, lw, bt, bw, counts, avg_dev, p, st, ht = ctx.keep
        L, dev = ctx.geom.L, cls[0].device
        z = _zero1(dev)
        gin = torch.cat([z if g is None else g.detach().reshape(1).to(torch.float32) for g in gs])
        gp, gst = LevelPtrs(), _lib.LevelPixStrides()
        cl = torch.channels_last
        g_cls = [torch.empty(t.shape, dtype=torch.float32, device=dev, memory_format=cl) for t in cls]
        if ctx.fused:
            # one gradient tensor per level shaped like the wider tensor; channels beyond reg | iou
            # (alignment padding) get zero gradient
            # (_shared_base guarantees reg = channels [0, 4A), iou right behind: with that STATED in the cfg the
            # backward kernel writes the zero gradient of the padding channels itself, ia_head_loss_bwd_nhwc;
            # rows with more than 64 padding channels are cleared here and the flag stays off)
            g_base = [torch.empty(b.shape, dtype=torch.float32, device=dev, memory_format=cl)
                      for b in bases]
            in_kernel = all(b.shape[1] - (reg[l].shape[1] + iou[l].shape[1]) <= 64 for l, b in enumerate(bases))
            ctx.cfg.grad_rows_start_at_reg = 1 if in_kernel else 0
            if not in_kernel:
                for gb in g_base:
                    gb.zero_()
            for l in range(L):
                gp.reg[l] = g_base[l].data_ptr() + (reg[l].data_ptr() - bases[l].data_ptr())
                gp.iou[l] = g_base[l].data_ptr() + (iou[l].data_ptr() - bases[l].data_ptr())
                gst.reg[l] = gst.iou[l] = bases[l].shape[1]
            tail = g_base
        else:
            g_reg = [torch.empty(t.shape, dtype=torch.float32, device=dev, memory_format=cl) for t in reg]
            g_iou = [torch.empty(t.shape, dtype=torch.float32, device=dev, memory_format=cl) for t in iou]
            for l in range(L):
                gp.reg[l], gp.iou[l] = g_reg[l].data_ptr(), g_iou[l].data_ptr()
                gst.reg[l], gst.iou[l] = reg[l].shape[1], iou[l].shape[1]
            tail = g_reg + g_iou
            ctx.cfg.grad_rows_start_at_reg = 0
        for l in range(L):
            gp.cls[l], gst.cls[l] = g_cls[l].data_ptr(), cls[l].shape[1]
        _lib.check(_lib.lib().ia_head_loss_bwd_nhwc(ctx.geom.ref(), C.byref(p), C.byref(st), ctx.B,
                                                    C.byref(ht), C.byref(ctx.cfg), _ptr(ctx.res),
                                                    _ptr(gin), C.byref(gp), C.byref(gst), _stream()),
                   'ia_head_loss_bwd_nhwc')
        return (None, None, None, None) + tuple(g_cls) + tuple(tail)


def _nhwc_route(geom, cls, reg, iou):
    """-> (views or None, inputs) when every head output is channels-last-like fp32, else None"""
    if geom.C % 4 or geom.A * (geom.C // 4) > 8192 \
            or any(_pix_stride(t) is None for t in list(cls) + list(reg) + list(iou)):
        return None
    if any((_pix_stride(t) % 4) for t in list(cls) + list(reg)) \
            or any(t.data_ptr() % 16 for t in list(cls) + list(reg)):
        return None
    bases = [_shared_base(r, i) for r, i in zip(reg, iou)]
    if all(b is not None for b in bases):
        return (list(reg), list(iou)), list(cls) + bases
    return None, list(cls) + list(reg) + list(iou)


def head_loss(geom, cls, reg, iou, labels, label_weights, bbox_targets, bbox_weights, counts=None,
              avg_factor=None, gamma=2.0, alpha=0.25, loss_weight_cls=1.0, beta=0.11,
              loss_weight_bbox=1.0, attach_iou_target=True, exact_large_logits=False,
              channels_last=None):
    """FocalLoss(gamma=2) + SmoothL1Loss + IoU BCE of every pyramid level in one autograd node.
    channels_last: None = the channels-last kernels when every head output is channels-last(-like)
    fp32 (the training head's outputs are), else the NCHW kernels; True / False force a route.
    Normaliser: `counts` ((B,2) of anchor_targets: sum_b max(n_pos_b, 1), stays on the device), else
    `avg_factor` (python number or device scalar).  -> dict of three LevelLosses lists."""
    if counts is None and avg_factor is None:
        raise ValueError('head_loss needs counts or avg_factor')
    if float(gamma) != 2.0:
        raise ValueError('the all-levels kernel is specialised for gamma = 2')
    L = geom.L
    cfg = (float(gamma), float(alpha), float(loss_weight_cls), float(beta), float(loss_weight_bbox),
           int(bool(attach_iou_target)), int(bool(exact_large_logits)))
    targets = (list(labels), list(label_weights), list(bbox_targets), list(bbox_weights), counts,
               avg_factor)
    route = _nhwc_route(geom, cls, reg, iou) if channels_last is not False else None
    if route is not None:
        flat = _HeadLossNhwcFn.apply(geom, targets, cfg, route[0], *route[1])
    elif channels_last is True:
        raise ValueError('channels_last=True needs channels-last fp32 head outputs, C % 4 == 0')
    else:
        flat = _HeadLossFn.apply(geom, targets, cfg, *(list(cls) + list(reg) + list(iou)))
    out = {}
    for k, name in enumerate(('loss_cls', 'loss_bbox', 'losses_iou')):
        lst = LevelLosses(flat[k * L:(k + 1) * L])
        lst.total = flat[3 * L + k]
        out[name] = lst
    return out


_FOCAL_OP_DTYPES = {torch.float32: _lib.IA_F32, torch.bfloat16: _lib.IA_BF16, torch.float16: _lib.IA_F16,
                    torch.float64: _lib.IA_F64}


class _SigmoidFocalLossOpFn(torch.autograd.Function):
    """the reference's mmdet.ops.sigmoid_focal_loss op (integer targets, no weights) in the
    logits' own storage type -- float, double, half like the reference's dispatch
    (sigmoid_focal_loss_cuda.cu:128,166), plus bf16: no fp32 copies around the kernel."""

    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        _require_gpu(logits, 'input')
        if logits.dim() != 2:
            raise RuntimeError('logits should be NxClass')
        if logits.dtype not in _FOCAL_OP_DTYPES:
            raise TypeError('sigmoid_focal_loss: floating-point logits expected, got %s' % logits.dtype)
        x = logits.contiguous()
        t = targets.contiguous().to(torch.int64)
        dt = _FOCAL_OP_DTYPES[x.dtype]
        out = torch.empty_like(x)
        _lib.check(_lib.lib().ia_sigmoid_focal_loss_fwd_dt(_ptr(x), dt, _ptr(t), x.shape[0], x.shape[1],
                                                           float(gamma), float(alpha), _ptr(out),
                                                           _stream()), 'ia_sigmoid_focal_loss_fwd_dt')
        ctx.save_for_backward(x, t)
        ctx.cfg = (float(gamma), float(alpha), dt)
        return out

    @staticmethod
    def backward(ctx, d_loss):
        x, t = ctx.saved_tensors
        gamma, alpha, dt = ctx.cfg
        d = d_loss.contiguous().to(x.dtype)
        out = torch.empty_like(x)
        _lib.check(_lib.lib().ia_sigmoid_focal_loss_bwd_dt(_ptr(x), dt, _ptr(t), _ptr(d), x.shape[0],
                                                           x.shape[1], gamma, alpha, _ptr(out),
                                                           _stream()), 'ia_sigmoid_focal_loss_bwd_dt')
        return out, None, None, None


def sigmoid_focal_loss_elementwise(logits, targets, gamma=2.0, alpha=0.25):
    return _SigmoidFocalLossOpFn.apply(logits, targets, gamma, alpha)
