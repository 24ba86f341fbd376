"""GPU: robustness of the C-ABI entries.

  * the fused row-max + top-k-filter launch with its waits forced to give up at once: the call
    must still return the oracle's detections (k_sel_final falls back to the complete row maxima),
    report the fallback in the status words, and leave the workspace ready for a normal call;
  * canary guard regions (256 bytes of 0xA5 in front of and behind) around every workspace a
    `*_workspace_bytes` query sizes: no entry point writes outside what it asked for."""
import numpy as np
import pytest
import torch

import gpu_util as G
import synth

pytestmark = pytest.mark.gpu
GUARD = 256


@pytest.fixture
def ops():
    from iouaware import ops as o
    return o


def _heads(seed, B, ph, pw, kind, layout):
    cls, reg, iou = synth.head_outputs(seed, B, ph, pw, kind)
    dev = [G.to_dev(x) for x in (cls, reg, iou)]
    if layout:
        dev = [[t.contiguous(memory_format=torch.channels_last) for t in x] for x in dev]
    return (cls, reg, iou), dev


@pytest.mark.parametrize('kind', ['A', 'B'])
def test_fused_launch_timeout_falls_back_to_the_dense_selection(ops, oracle_lib, kind):
    ph, pw, B = 800, 1344, 2
    geom, base = G.geometry(ph, pw, 1000)
    host, dev = _heads(31, B, ph, pw, kind, 1)
    shapes, sfs = [(800, 1333, 3)] * B, [1.0] * B
    g, _, ws = ops.state_workspace_for(geom, *dev)
    assert g.layout == 1
    before = ops.get_bboxes_status(g, B, ws)
    ops.fused_spin_limit(0)                       # every wait of the fused launch gives up at once
    try:
        forced = [t.cpu().numpy() for t in ops.get_bboxes(geom, *dev, shapes, sfs, True, 0.05, 0.5, 100)]
        mid = ops.get_bboxes_status(g, B, ws)
    finally:
        ops.fused_spin_limit(-1)
    normal = [t.cpu().numpy() for t in ops.get_bboxes(geom, *dev, shapes, sfs, True, 0.05, 0.5, 100)]
    after = ops.get_bboxes_status(g, B, ws)
    assert mid[1] == before[1] + 1 and mid[0] != 0, (before, mid)      # the forced call fell back, and said so
    assert after == mid, (mid, after)                                   # the normal call did not
    for got in (forced, normal):
        dets, labels, rows, num = got
        for b in range(B):
            o = oracle_lib.get_bboxes_single([x[b] for x in host[0]], [x[b] for x in host[1]],
                                             [x[b] for x in host[2]], synth.STRIDES, base, (800, 1333),
                                             1.0, True, 1000, 0.05, 0.5, 100)
            n = int(num[b])
            assert n == o['num_det'] and n > 0
            assert np.array_equal(rows[b, :n], o['det_rows'])
            assert np.array_equal(labels[b, :n], o['det_labels'])
            assert G.same_bits(dets[b, :n], o['det_bboxes'])


class _Guarded(object):
    """hands out workspaces with guard regions and checks them afterwards"""

    def __init__(self):
        self.bufs = []

    def make(self, device, nbytes, zero):
        n = int(nbytes)
        pad = (-n) % 256
        buf = torch.full((GUARD + n + pad + GUARD,), 0xA5, dtype=torch.uint8, device=device)
        if zero:
            buf[GUARD:GUARD + n].zero_()
        self.bufs.append((buf, n))
        return buf[GUARD:GUARD + n]

    def check(self):
        assert self.bufs, 'no workspace was requested'
        for buf, n in self.bufs:
            assert bool((buf[:GUARD] == 0xA5).all()), 'write in front of a %d-byte workspace' % n
            assert bool((buf[GUARD + n:] == 0xA5).all()), 'write behind a %d-byte workspace' % n


@pytest.fixture
def guarded(ops, monkeypatch):
    gd = _Guarded()
    monkeypatch.setattr(ops, '_state_workspace', lambda dev, nbytes, key: gd.make(dev, nbytes, True))
    monkeypatch.setattr(ops, '_workspace', lambda dev, nbytes: gd.make(dev, nbytes, False))
    monkeypatch.setattr(ops, '_own_workspace', lambda dev, nbytes: gd.make(dev, nbytes, False))
    monkeypatch.setattr(ops, '_col_buffer', lambda dev, nbytes: gd.make(dev, nbytes, False))
    monkeypatch.setattr(ops, 'select_workspace', lambda geom, batch, dev: gd.make(
        dev, max(int(ops._lib.lib().ia_select_topk_workspace_bytes(geom.ref(), int(batch))), 1), True))
    yield gd
    torch.cuda.synchronize()


@pytest.mark.parametrize('layout', [0, 1])
@pytest.mark.parametrize('B,ph,pw,kind', [(1, 256, 320, 'A'), (3, 224, 288, 'B'), (2, 800, 1344, 'B'),
                                          (8, 800, 1344, 'A')])
def test_get_bboxes_stays_inside_its_workspace(ops, guarded, layout, B, ph, pw, kind):
    """ia_get_bboxes_lazy, ia_get_bboxes (complete NMS) and the soft-NMS chain"""
    geom, _ = G.geometry(ph, pw, 1000)
    _, dev = _heads(7, B, ph, pw, kind, layout)
    shapes, sfs = [(ph, pw - 3, 3)] * B, [1.0] * B
    ops.get_bboxes(geom, *dev, shapes, sfs, True, 0.05, 0.5, 100)
    ops.get_bboxes(geom, *dev, shapes, sfs, True, 0.05, 0.5, 100, lazy=False)
    ops.get_bboxes(geom, *dev, shapes, sfs, True, 0.05, 0.5, 100, lazy_candidates=128)
    torch.cuda.synchronize()
    guarded.check()


def test_stage_entries_stay_inside_their_workspaces(ops, guarded):
    """the stage-wise entries: select workspace, NMS workspaces (complete, lazy, soft), ia_nms for a
    few sizes incl. the chunked path above 8192 boxes"""
    geom0, _ = G.geometry(800, 1344, 1000)
    B = 2
    _, dev = _heads(8, B, 800, 1344, 'B', 1)
    geom = ops.geometry_for(geom0, *dev)
    shapes, sfs = [(800, 1333, 3)] * B, [1.0] * B
    rm = ops.decode_fuse_rowmax(geom, *dev)
    idx = ops.select_topk(geom, rm)
    boxes, scores_t, best = ops.gather_decode(geom, *dev, idx, shapes, sfs, True)
    ops.multiclass_nms(boxes, scores_t, geom.R, 0.05, 0.5, 100, best_score=best)
    ops.multiclass_nms_lazy(boxes, scores_t, geom.R, 0.05, 0.5, 100, best_score=best)
    ops.multiclass_soft_nms(boxes, scores_t, geom.R, 0.05, 0.5, 100)
    g = torch.Generator(device='cuda').manual_seed(2)
    for nb in (1, 63, 4693, 8192, 8193, 20000):
        xy = torch.rand(nb, 2, device='cuda', generator=g) * 900
        wh = torch.rand(nb, 2, device='cuda', generator=g) * 120 + 4
        dets = torch.cat([xy, xy + wh, torch.rand(nb, 1, device='cuda', generator=g)], 1)
        ops.nms_indices(dets, 0.5)
    torch.cuda.synchronize()
    guarded.check()


def test_conv_entries_stay_inside_their_workspaces(ops, guarded):
    """library GEMM workspace (64 MiB request), im2col matrix, strided projection"""
    x = torch.randn(2, 64, 25, 42, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(9 * 64, 32, device='cuda')
    ops.conv3x3_im2col(x, w, None, stride=2, relu=True)
    ops.conv1x1_strided(x, torch.randn(64, 48, device='cuda'), None, None, stride=2)
    ops.linear_bias_act(x, torch.randn(64, 128, device='cuda'), torch.randn(128, device='cuda'), relu=True)
    torch.cuda.synchronize()
    guarded.check()


def test_head_loss_stays_inside_its_workspace(ops, guarded):
    B, ph, pw = 2, 256, 320
    geom, _ = G.geometry(ph, pw, -1)
    (cls, reg, iou), dev = _heads(9, B, ph, pw, 'A', 0)
    gts, gls = synth.train_targets(3, B, ph - 6, pw - 3, max_gt=6)
    gtb = [torch.from_numpy(x).cuda() for x in gts]
    gtl = [torch.from_numpy(x).cuda() for x in gls]
    labels, lw, bt, bw, counts = ops.anchor_targets(geom, gtb, gtl, [(ph, pw, 3)] * B, 0.5, 0.4, 0.0, -1)
    outs = [[t.clone().requires_grad_(True) for t in x] for x in dev]
    for cl in (False, None):
        losses = ops.head_loss(geom, *outs, labels, lw, bt, bw, counts=counts, channels_last=cl)
        sum(v.total for v in losses.values()).backward()
    torch.cuda.synchronize()
    guarded.check()


def test_stage_events_bracket_the_decode_stage_and_change_nothing(ops):
    """ia_profile_stage_events (bench.py's in-step roofline timing): the two events are recorded by
    every later get_bboxes call on its stream, the bracketed time is the decode stage's (tens of
    microseconds at this size, far below the whole call), the detections are unchanged, and
    (None, None) switches the hook off again."""
    ph, pw, B = 800, 1344, 2
    geom, _ = G.geometry(ph, pw, 1000)
    _, dev = _heads(37, B, ph, pw, 'A', 1)
    shapes, sfs = [(800, 1333, 3)] * B, [1.0] * B
    plain = [t.cpu().numpy() for t in ops.get_bboxes(geom, *dev, shapes, sfs, True, 0.05, 0.5, 100)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with pytest.raises(ValueError):
        ops.stage_events(e0, e1)                      # not created yet
    e0.record()
    e1.record()
    torch.cuda.synchronize()
    ops.stage_events(e0, e1)
    try:
        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0.record()
        timed = [t.cpu().numpy() for t in ops.get_bboxes(geom, *dev, shapes, sfs, True, 0.05, 0.5, 100)]
        w1.record()
        torch.cuda.synchronize()
        stage_ms, call_ms = e0.elapsed_time(e1), w0.elapsed_time(w1)
    finally:
        ops.stage_events(None, None)
    assert 0.0 < stage_ms < call_ms, (stage_ms, call_ms)
    for a, b in zip(plain, timed):
        assert np.array_equal(a, b)
    # switched off: a later call leaves the pair untouched
    ops.get_bboxes(geom, *dev, shapes, sfs, True, 0.05, 0.5, 100)
    torch.cuda.synchronize()
    assert e0.elapsed_time(e1) == stage_ms
