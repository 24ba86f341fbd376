"""GPU (MI355X): the HIP loss kernels (csrc/loss.hip) through the C-ABI / the
autograd wrappers, against the CPU oracle (smooth-L1 / IoU-BCE elementwise
gradients and IoU targets BIT-EXACT; focal within 1e-5 -- it uses the hardware
transcendental units; sums within 2e-6 relative: the reduction order differs)
and against the reference's own losses / autograd gradients in
tests/golden/losses_small.npz (1e-4)."""
import os

import numpy as np
import pytest
import torch

import synth
import gpu_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available()
    from iouaware import ops as o
    return o


@pytest.fixture(scope='module')
def fx(golden_dir):
    f = np.load(os.path.join(golden_dir, 'losses_small.npz'))
    ih, iw, ph, pw = [int(v) for v in f['img']]
    B = int(f['batch'])
    cls, reg, iou = synth.head_outputs(int(f['seed']), B, ph, pw, str(f['kind']))
    return f, cls, reg, iou, B, (ph, pw)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def test_focal_smoothl1_ioubce_vs_oracle_and_reference(ops, oracle_lib, fx):
    f, cls, reg, iou, B, (ph, pw) = fx
    geom, base = G.geometry(ph, pw, -1)
    avg = float(f['num_total_pos'])
    gs = np.float32(1.0 / avg)
    for l, (h, w) in enumerate(geom.featmap_sizes):
        n_l = h * w * synth.A
        labels = torch.from_numpy(f['labels_%d' % l]).cuda()
        lw = torch.from_numpy(f['label_weights_%d' % l]).cuda()
        bt = torch.from_numpy(f['bbox_targets_%d' % l]).cuda().reshape(B, n_l, 4)
        bw = torch.from_numpy(f['bbox_weights_%d' % l]).cuda().reshape(B, n_l, 4)
        c = torch.from_numpy(cls[l]).cuda().requires_grad_(True)
        r = torch.from_numpy(reg[l]).cuda().requires_grad_(True)
        i = torch.from_numpy(iou[l]).cuda().requires_grad_(True)
        # the loss exactly as the head composes it: sum * (loss_weight / avg_factor)
        lc = ops.focal_loss_sum(c, labels, lw, synth.A, 2.0, 0.25) * float(gs)
        lb = ops.smooth_l1_sum(r, bt, bw, synth.A, 0.11) * float(gs)
        li = ops.iou_bce_sum(r, i, bt, bw, geom, l, True) * float(gs)
        (lc + lb + li).sum().backward()
        torch.cuda.synchronize()
        # --- vs oracle
        so, go = oracle_lib.focal_loss(cls[l], f['labels_%d' % l], f['label_weights_%d' % l],
                                       synth.A, 2.0, 0.25, gscale=float(gs))
        # focal runs on the hardware exp/log/rcp units (HBM-bound streaming kernel): not
        # bit-identical to the oracle's software math, but far inside the 1e-4 bar
        assert rel(float(lc) / float(gs), so) < 2e-6
        gd = c.grad.cpu().numpy().astype(np.float64)
        assert (np.abs(gd - go) <= 1e-5 * np.abs(go) + 1e-6 * np.abs(go).max()).all(), \
            'focal grad level %d' % l
        so, go = oracle_lib.smooth_l1(reg[l], f['bbox_targets_%d' % l], f['bbox_weights_%d' % l],
                                      synth.A, 0.11, gscale=float(gs))
        assert abs(float(lb) / float(gs) - so) <= 1e-6 * max(abs(so), 1e-6)
        so2, tgt, g_iou, g_box = oracle_lib.iou_bce(reg[l], iou[l], f['bbox_targets_%d' % l],
                                                    f['bbox_weights_%d' % l], base[l],
                                                    synth.STRIDES[l], gscale=float(gs))
        assert abs(float(li) / float(gs) - so2) <= 1e-6 * max(abs(so2), 1e-6)
        assert G.same_bits(i.grad.cpu().numpy(), g_iou), 'iou grad level %d' % l
        # bbox_pred grad = smooth-L1 part + IoU-target part, summed by autograd in fp32
        assert G.same_bits(r.grad.cpu().numpy(), go + g_box), 'reg grad level %d' % l
        t_dev, _ = ops.iou_targets(r.detach(), i.detach(), bt, bw, geom, l)
        assert G.same_bits(t_dev.cpu().numpy(), tgt)
        # --- vs the reference (golden)
        assert rel(float(lc), f['loss_cls'][l]) < 1e-4
        assert abs(float(lb) - f['loss_bbox'][l]) <= 1e-4 * max(f['loss_bbox'][l], 1e-6)
        assert abs(float(li) - f['losses_iou'][l]) <= 1e-4 * max(f['losses_iou'][l], 1e-6)
        for key, g, mode, tol in (('g_cls_%d' % l, c.grad, 'attached', 1e-4),
                                  ('g_iou_%d' % l, i.grad, 'attached', 1e-4),
                                  ('g_reg_%d' % l, r.grad, 'attached', 2e-4)):
            idx = f[key + '_idx']
            want = f['%s_%s' % (key, mode)].astype(np.float64)
            got = g.cpu().numpy().reshape(-1)[idx].astype(np.float64)
            assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-30), key


def test_detached_iou_target_variant(ops, oracle_lib, fx):
    f, cls, reg, iou, B, (ph, pw) = fx
    geom, base = G.geometry(ph, pw, -1)
    l = 0
    h, w = geom.featmap_sizes[l]
    n_l = h * w * synth.A
    bt = torch.from_numpy(f['bbox_targets_%d' % l]).cuda().reshape(B, n_l, 4)
    bw = torch.from_numpy(f['bbox_weights_%d' % l]).cuda().reshape(B, n_l, 4)
    r = torch.from_numpy(reg[l]).cuda().requires_grad_(True)
    i = torch.from_numpy(iou[l]).cuda().requires_grad_(True)
    ops.iou_bce_sum(r, i, bt, bw, geom, l, False).sum().backward()
    assert r.grad is None and i.grad is not None


def test_head_loss_end_to_end_vs_reference(ops, fx):
    """IoUawareRetinaHead.loss (targets in torch on the GPU + HIP losses) reproduces the
    reference's loss dict for the same head outputs and gt boxes."""
    from iouaware.config import ConfigDict
    from iouaware.head import IoUawareRetinaHead
    from test_host_targets import HEAD_KW, TRAIN_CFG
    f, cls, reg, iou, B, (ph, pw) = fx
    ih, iw = int(f['img'][0]), int(f['img'][1])
    head = IoUawareRetinaHead(**HEAD_KW).cuda()
    metas = [synth.img_meta(ih, iw, ph, pw) for _ in range(B)]
    gts = [torch.from_numpy(f['gt_bboxes_%d' % b]).cuda() for b in range(B)]
    gls = [torch.from_numpy(f['gt_labels_%d' % b]).cuda() for b in range(B)]
    c = [t.requires_grad_(True) for t in G.to_dev(cls)]
    r = [t.requires_grad_(True) for t in G.to_dev(reg)]
    i = [t.requires_grad_(True) for t in G.to_dev(iou)]
    losses = head.loss(c, r, i, gts, gls, metas, TRAIN_CFG)
    assert sorted(losses) == ['loss_bbox', 'loss_cls', 'losses_iou']     # key spelled as :387
    for k in losses:
        got = np.array([float(x) for x in losses[k]])
        assert all(x.shape == (1,) for x in losses[k])
        assert np.all(np.abs(got - f[k]) <= 1e-4 * np.maximum(np.abs(f[k]), 1e-6)), k
    sum(sum(v) for v in losses.values()).backward()
    for l in range(5):
        for key, g in (('g_cls_%d' % l, c[l].grad), ('g_reg_%d' % l, r[l].grad),
                       ('g_iou_%d' % l, i[l].grad)):
            idx = f[key + '_idx']
            want = f[key + '_attached'].astype(np.float64)
            got = g.cpu().numpy().reshape(-1)[idx].astype(np.float64)
            assert np.abs(got - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-30), key


def test_sigmoid_focal_loss_op(ops, oracle_lib):
    """mmdet.ops.sigmoid_focal_loss semantics (integer targets, elementwise), bit-exact vs oracle."""
    from iouaware.focal_op import sigmoid_focal_loss, SigmoidFocalLoss
    rs = np.random.RandomState(4)
    N, Cn = 3000, 80
    x = (rs.standard_normal((N, Cn)) * 4).astype(np.float32)
    t = rs.randint(-1, Cn + 1, N).astype(np.int64)
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    td = torch.from_numpy(t).cuda()
    out = sigmoid_focal_loss(xd, td, 2.0, 0.25, 'none')
    assert G.same_bits(out.detach().cpu().numpy(), oracle_lib.focal_loss_op(x, t, 2.0, 0.25))
    dl = rs.uniform(0.5, 1.5, (N, Cn)).astype(np.float32)
    out.backward(torch.from_numpy(dl).cuda())
    assert G.same_bits(xd.grad.cpu().numpy(), oracle_lib.focal_loss_op(x, t, 2.0, 0.25, dl))
    m = SigmoidFocalLoss(2.0, 0.25)
    # the reference module reduces with the function's default 'mean' (modules/...:17-19)
    assert abs(float(m(xd.detach(), td)) - float(out.mean())) <= 1e-4 * abs(float(out.mean()))
    with pytest.raises(AssertionError):
        m(xd.detach().cpu(), td.cpu())


def test_loss_module_signatures_on_permuted_inputs(ops, oracle_lib, fx):
    """FocalLoss / SmoothL1Loss called the reference way: (N,C) scores, one-hot labels,
    expanded weights (focal_loss.py:23-35, smooth_l1_loss.py:14-18)."""
    from iouaware.losses import FocalLoss, SmoothL1Loss
    from iouaware.targets import expand_binary_labels
    f, cls, reg, iou, B, (ph, pw) = fx
    l = 2
    labels = torch.from_numpy(f['labels_%d' % l]).reshape(-1).cuda()
    lw = torch.from_numpy(f['label_weights_%d' % l]).reshape(-1).cuda()
    onehot, wexp = expand_binary_labels(labels, lw, 80)
    score = torch.from_numpy(cls[l]).cuda().permute(0, 2, 3, 1).reshape(-1, 80)
    avg = float(f['num_total_pos'])
    got = FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25)(score, onehot, wexp, avg_factor=avg)
    assert got.shape == (1,) and rel(float(got), f['loss_cls'][l]) < 1e-4
    pred = torch.from_numpy(reg[0]).cuda().permute(0, 2, 3, 1).reshape(-1, 4)
    bt = torch.from_numpy(f['bbox_targets_0']).cuda().reshape(-1, 4)
    bw = torch.from_numpy(f['bbox_weights_0']).cuda().reshape(-1, 4)
    got = SmoothL1Loss(beta=0.11)(pred, bt, bw, avg_factor=avg)
    assert abs(float(got) - f['loss_bbox'][0]) <= 1e-4 * f['loss_bbox'][0]


# ------------------------------------------------------------------ IoU-balanced variants (8f.4)
def test_iou_balanced_losses_vs_oracle(ops, oracle_lib, fx):
    f, cls, reg, iou, B, (ph, pw) = fx
    geom, base = G.geometry(ph, pw, -1)
    gs = float(np.float32(1.0 / float(f['num_total_pos'])))
    for l, (h, w) in enumerate(geom.featmap_sizes):
        n_l = h * w * synth.A
        labels = torch.from_numpy(f['labels_%d' % l]).cuda()
        lw = torch.from_numpy(f['label_weights_%d' % l]).cuda()
        bt = torch.from_numpy(f['bbox_targets_%d' % l]).cuda().reshape(B, n_l, 4)
        bw = torch.from_numpy(f['bbox_weights_%d' % l]).cuda().reshape(B, n_l, 4)
        c = torch.from_numpy(cls[l]).cuda().requires_grad_(True)
        r = torch.from_numpy(reg[l]).cuda().requires_grad_(True)
        i = torch.from_numpy(iou[l]).cuda()
        _, tgt = ops.iou_bce_sum(r.detach(), i, bt, bw, geom, l, True, return_iou=True)
        _, otgt, _, _ = oracle_lib.iou_bce(reg[l], iou[l], f['bbox_targets_%d' % l],
                                           f['bbox_weights_%d' % l], base[l], synth.STRIDES[l])
        assert G.same_bits(tgt.cpu().numpy(), otgt)
        lc = ops.focal_loss_balanced_sum(c, labels, lw, tgt, synth.A, 2.0, 0.25, 1.5) * gs
        lb = ops.smooth_l1_balanced_sum(r, bt, bw, tgt, synth.A, 0.11, 1.5) * gs
        (lc + lb).sum().backward()
        so, go, sums = oracle_lib.focal_loss_balanced(cls[l], f['labels_%d' % l],
                                                      f['label_weights_%d' % l], otgt, synth.A,
                                                      2.0, 0.25, 1.5, gscale=gs)
        assert rel(float(lc) / gs, so) < 1e-5, (l, float(lc) / gs, so)
        gd = c.grad.cpu().numpy().astype(np.float64)
        assert (np.abs(gd - go) <= 2e-5 * np.abs(go) + 2e-6 * np.abs(go).max()).all(), l
        sb, gb = oracle_lib.smooth_l1_balanced(reg[l], f['bbox_targets_%d' % l],
                                               f['bbox_weights_%d' % l], otgt, synth.A, 0.11, 1.5,
                                               gscale=gs)
        assert abs(float(lb) / gs - sb) <= 1e-6 * max(abs(sb), 1e-6)
        assert G.same_bits(r.grad.cpu().numpy(), gb), 'balanced smooth-L1 grad level %d' % l


def test_head_loss_iou_balanced_vs_reference(ops, fx, golden_dir):
    """loss_cls.type='IOUbalancedSigmoidFocalLoss', loss_bbox.type='IoUbalancedSmoothL1Loss'
    through IoUawareRetinaHead.loss against the reference's own output."""
    from iouaware.head import IoUawareRetinaHead
    from test_host_targets import HEAD_KW, TRAIN_CFG
    f, cls, reg, iou, B, (ph, pw) = fx
    fb = np.load(os.path.join(golden_dir, 'losses_balanced.npz'))
    kw = dict(HEAD_KW)
    kw['loss_cls'] = dict(type='IOUbalancedSigmoidFocalLoss', use_sigmoid=True, gamma=2.0,
                          alpha=0.25, eta=1.5, loss_weight=1.0)
    kw['loss_bbox'] = dict(type='IoUbalancedSmoothL1Loss', beta=0.11, delta=1.5, loss_weight=3.049)
    head = IoUawareRetinaHead(**kw).cuda()
    assert head.IoU_balanced_Cls and head.IoU_balanced_Loc and not head.sampling
    ih, iw = int(f['img'][0]), int(f['img'][1])
    metas = [synth.img_meta(ih, iw, ph, pw) for _ in range(B)]
    gts = [torch.from_numpy(f['gt_bboxes_%d' % b]).cuda() for b in range(B)]
    gls = [torch.from_numpy(f['gt_labels_%d' % b]).cuda() for b in range(B)]
    c = [t.requires_grad_(True) for t in G.to_dev(cls)]
    r = [t.requires_grad_(True) for t in G.to_dev(reg)]
    i = [t.requires_grad_(True) for t in G.to_dev(iou)]
    losses = head.loss(c, r, i, gts, gls, metas, TRAIN_CFG)
    for k in losses:
        got = np.array([float(x) for x in losses[k]])
        assert np.all(np.abs(got - fb[k]) <= 1e-4 * np.maximum(np.abs(fb[k]), 1e-6)), (k, got, fb[k])
    sum(sum(v) for v in losses.values()).backward()
    for l in range(5):
        for key, g in (('g_cls_%d' % l, c[l].grad), ('g_reg_%d' % l, r[l].grad),
                       ('g_iou_%d' % l, i[l].grad)):
            want = fb[key].astype(np.float64)
            got = g.cpu().numpy().reshape(-1)[fb[key + '_idx']].astype(np.float64)
            assert np.abs(got - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-30), key


# ------------------------------------------------------------------ all levels in one node
def _head_and_inputs(fx, dtype=torch.float32):
    from iouaware.head import IoUawareRetinaHead
    from test_host_targets import HEAD_KW
    f, cls, reg, iou, B, (ph, pw) = fx
    ih, iw = int(f['img'][0]), int(f['img'][1])
    head = IoUawareRetinaHead(**HEAD_KW).cuda()
    metas = [synth.img_meta(ih, iw, ph, pw) for _ in range(B)]
    gts = [torch.from_numpy(f['gt_bboxes_%d' % b]).cuda() for b in range(B)]
    gls = [torch.from_numpy(f['gt_labels_%d' % b]).cuda() for b in range(B)]
    mk = lambda xs: [t.requires_grad_(True) for t in G.to_dev(xs, dtype)]     # noqa: E731
    return head, metas, gts, gls, mk(cls), mk(reg), mk(iou)


@pytest.mark.parametrize('attach', [True, False])
def test_all_levels_loss_node_equals_per_level_kernels(ops, fx, attach):
    """csrc/headloss.hip (3 + 2 launches) against the per-level kernels of csrc/loss.hip on the
    same targets: losses to 1e-6, gradients to 1e-6 of their scale (focal: other operation
    order, hardware transcendentals on both sides)."""
    from test_host_targets import TRAIN_CFG
    outs = []
    for fuse in (True, False):
        head, metas, gts, gls, c, r, i = _head_and_inputs(fx)
        head.fuse_levels, head.attach_iou_target = fuse, attach
        losses = head.loss(c, r, i, gts, gls, metas, TRAIN_CFG)
        assert isinstance(losses['loss_cls'], ops.LevelLosses) == fuse
        # upstream gradients that differ per loss and level
        w = torch.arange(1, 16, device='cuda', dtype=torch.float32).reshape(3, 5) * 0.25
        total = sum(w[k, l] * losses[key][l] for k, key in
                    enumerate(('loss_cls', 'loss_bbox', 'losses_iou')) for l in range(5))
        total.sum().backward()
        outs.append((losses, [t.grad for t in c], [t.grad for t in r], [t.grad for t in i]))
    (la, ca, ra, ia_), (lb, cb, rb, ib) = outs
    for k in la:
        for x, y in zip(la[k], lb[k]):
            assert x.shape == (1,) and rel(float(x), float(y)) < 1e-6, k
        assert rel(float(la[k].total), sum(float(v) for v in lb[k])) < 1e-6
    for name, xs, ys in (('cls', ca, cb), ('reg', ra, rb), ('iou', ia_, ib)):
        for l, (x, y) in enumerate(zip(xs, ys)):
            scale = float(y.abs().max())
            assert float((x - y).abs().max()) <= 1e-6 * max(scale, 1e-30), (name, l)
            assert x.shape == y.shape and x.dtype == y.dtype


def test_all_levels_loss_node_total_path_and_reference(ops, fx):
    """parse_losses adds the three `.total` tensors; gradients through them equal the reference's
    autograd gradients (tests/golden/losses_small.npz, 1e-4 / 2e-4 like the per-level test)."""
    from iouaware.train import parse_losses
    from test_host_targets import TRAIN_CFG
    f = fx[0]
    head, metas, gts, gls, c, r, i = _head_and_inputs(fx)
    losses = head.loss(c, r, i, gts, gls, metas, TRAIN_CFG)
    loss, log_vars = parse_losses(losses)
    want = float(f['loss_cls'].sum() + f['loss_bbox'].sum() + f['losses_iou'].sum())
    assert rel(float(loss), want) < 1e-4
    for k in ('loss_cls', 'loss_bbox', 'losses_iou'):
        assert rel(float(log_vars[k]), float(f[k].sum())) < 1e-4
    loss.backward()
    for l in range(5):
        for key, g in (('g_cls_%d' % l, c[l].grad), ('g_reg_%d' % l, r[l].grad),
                       ('g_iou_%d' % l, i[l].grad)):
            idx = f[key + '_idx']
            want = f[key + '_attached'].astype(np.float64)
            got = g.cpu().numpy().reshape(-1)[idx].astype(np.float64)
            assert np.abs(got - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-30), key


def test_all_levels_loss_node_torch_targets_and_bf16(ops, fx):
    """host-number normaliser (torch target path) and bf16 head outputs"""
    from iouaware.targets import anchor_target
    from test_host_targets import TRAIN_CFG
    f = fx[0]
    head, metas, gts, gls, c, r, i = _head_and_inputs(fx)
    sizes = [tuple(t.shape[-2:]) for t in c]
    anchors, flags = head.get_anchors(sizes, metas, device='cuda')
    t = anchor_target(anchors, flags, gts, metas, head.target_means, head.target_stds, TRAIN_CFG,
                      gt_labels_list=gls, label_channels=80, sampling=False)
    geom = head.geometry(sizes, -1)
    out = ops.head_loss(geom, c, r, i, t[0], t[1], t[2], t[3], avg_factor=t[4])
    for k in ('loss_cls', 'loss_bbox', 'losses_iou'):
        got = np.array([float(x) for x in out[k]])
        assert np.all(np.abs(got - f[k]) <= 1e-4 * np.maximum(np.abs(f[k]), 1e-6)), k
    out2 = ops.head_loss(geom, c, r, i, t[0], t[1], t[2], t[3],
                         avg_factor=torch.tensor([float(t[4])], device='cuda'))
    assert all(float(a) == float(b) for k in out for a, b in zip(out[k], out2[k]))
    # bf16 logits: the kernels read bf16, accumulate in fp32 / fp64, return bf16 gradients
    hb, _, _, _, cb, rb, ib = _head_and_inputs(fx, torch.bfloat16)
    cf = [x.detach().float().requires_grad_(True) for x in cb]
    rf = [x.detach().float().requires_grad_(True) for x in rb]
    if_ = [x.detach().float().requires_grad_(True) for x in ib]
    a = ops.head_loss(geom, cb, rb, ib, t[0], t[1], t[2], t[3], avg_factor=t[4])
    b = ops.head_loss(geom, cf, rf, if_, t[0], t[1], t[2], t[3], avg_factor=t[4])
    sum(v.total for v in a.values()).sum().backward()
    sum(v.total for v in b.values()).sum().backward()
    for k in a:
        for x, y in zip(a[k], b[k]):
            assert rel(float(x), float(y)) < 1e-6          # same (bf16-exact) inputs
    for xs, ys in ((cb, cf), (rb, rf), (ib, if_)):
        for x, y in zip(xs, ys):
            assert x.grad.dtype == torch.bfloat16
            assert torch.equal(x.grad, y.grad.to(torch.bfloat16))


def test_all_levels_focal_full_size_vs_oracle(ops, oracle_lib):
    """800x1344, batch 2: the all-levels focal kernel against the fp64-summing oracle per level
    (1e-5 on the sums, 1e-5 of the gradient scale), incl. ignored anchors and every class id"""
    ph, pw, B = 800, 1344, 2
    geom, base = G.geometry(ph, pw, -1)
    cls, reg, iou = synth.head_outputs(31, B, ph, pw, 'A')
    rs = np.random.RandomState(5)
    labels, lw, bt, bw = [], [], [], []
    for (h, w) in geom.featmap_sizes:
        n = h * w * synth.A
        lab = np.zeros((B, n), np.int64)
        pos = rs.rand(B, n) < 0.004
        lab[pos] = rs.randint(1, 81, int(pos.sum()))
        wgt = (rs.rand(B, n) > 0.05).astype(np.float32)          # 5 % ignored
        labels.append(lab); lw.append(wgt)
        bt.append((rs.standard_normal((B, n, 4)) * 0.2 * pos[..., None]).astype(np.float32))
        bw.append(np.repeat(pos[..., None].astype(np.float32), 4, -1))
    dev = lambda xs: [torch.from_numpy(x).cuda() for x in xs]    # noqa: E731
    c = [t.requires_grad_(True) for t in G.to_dev(cls)]
    r = [t.requires_grad_(True) for t in G.to_dev(reg)]
    i = [t.requires_grad_(True) for t in G.to_dev(iou)]
    avg = 37.0
    out = ops.head_loss(geom, c, r, i, dev(labels), dev(lw), dev(bt), dev(bw), avg_factor=avg)
    sum(v.total for v in out.values()).sum().backward()
    for l in range(geom.L):
        so, go = oracle_lib.focal_loss(cls[l], labels[l], lw[l], synth.A, 2.0, 0.25,
                                       gscale=1.0 / avg)
        assert rel(float(out['loss_cls'][l]), so / avg) < 1e-5, l
        g = c[l].grad.cpu().numpy()
        assert np.abs(g - go).max() <= 1e-5 * np.abs(go).max(), l
        s1, g1 = oracle_lib.smooth_l1(reg[l], bt[l], bw[l], synth.A, 0.11, gscale=1.0 / avg)
        s2, tgt, g_iou, g_box = oracle_lib.iou_bce(reg[l], iou[l], bt[l], bw[l], base[l],
                                                   synth.STRIDES[l], gscale=1.0 / avg)
        assert rel(float(out['loss_bbox'][l]), s1 / avg) < 1e-5, l
        assert rel(float(out['losses_iou'][l]), s2 / avg) < 1e-5, l
        gr = r[l].grad.cpu().numpy()
        assert np.abs(gr - (g1 + g_box)).max() <= 1e-6 * max(np.abs(g1 + g_box).max(), 1e-30), l
        gi = i[l].grad.cpu().numpy()
        assert np.abs(gi - g_iou).max() <= 1e-6 * max(np.abs(g_iou).max(), 1e-30), l


def _cl(ts):
    return [t.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True) for t in ts]


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('attach', [True, False])
def test_channels_last_loss_kernels_equal_nchw_kernels(ops, fx, fused, attach):
    """k_focal_nhwc / k_box_nhwc (channels-last head outputs, no layout copies) against the NCHW
    all-levels kernels on the same values: losses 1e-6, gradients 1e-6 of their scale.  fused:
    reg / iou are channel slices of one 48-channel tensor, as the training head produces them --
    the gradient arrives as ONE tensor of that shape with zeros in the padding channels."""
    from test_host_targets import TRAIN_CFG
    head, metas, gts, gls, c, r, i = _head_and_inputs(fx)
    head.attach_iou_target = attach
    w = torch.arange(1, 16, device='cuda', dtype=torch.float32).reshape(3, 5) * 0.25

    def run(c_, r_, i_):
        losses = head.loss(c_, r_, i_, gts, gls, metas, TRAIN_CFG)
        total = sum(w[k, l] * losses[key][l] for k, key in
                    enumerate(('loss_cls', 'loss_bbox', 'losses_iou')) for l in range(5))
        total.sum().backward()
        return losses
    la = run(c, r, i)                                       # NCHW route (contiguous inputs)
    c2 = _cl(c)
    if fused:
        bases = []
        for rr, ii in zip(r, i):
            pad = torch.randn(rr.shape[0], 3, *rr.shape[2:], device='cuda')
            bases.append(torch.cat([rr.detach(), ii.detach(), pad], 1)
                         .contiguous(memory_format=torch.channels_last).requires_grad_(True))
        n_reg, n_iou = r[0].shape[1], i[0].shape[1]
        r2 = [b[:, :n_reg] for b in bases]
        i2 = [b[:, n_reg:n_reg + n_iou] for b in bases]
        assert ops._nhwc_route(head.geometry([tuple(t.shape[-2:]) for t in c], -1), c2, r2, i2)[0] is not None
    else:
        r2, i2 = _cl(r), _cl(i)
    lb = run(c2, r2, i2)
    for k in la:
        for x, y in zip(la[k], lb[k]):
            assert rel(float(x), float(y)) < 1e-6, k
    def close(x, y):
        return x.shape == y.shape and float((x - y).abs().max()) <= 1e-6 * max(float(y.abs().max()), 1e-30)
    for l in range(5):
        assert close(c2[l].grad, c[l].grad), l
        assert c2[l].grad.is_contiguous(memory_format=torch.channels_last)
        if fused:
            g = bases[l].grad
            assert g.is_contiguous(memory_format=torch.channels_last)
            assert close(g[:, :n_reg], r[l].grad) and close(g[:, n_reg:n_reg + n_iou], i[l].grad), l
            assert float(g[:, n_reg + n_iou:].abs().max()) == 0.0
        else:
            assert close(r2[l].grad, r[l].grad) and close(i2[l].grad, i[l].grad), l


def test_channels_last_focal_full_size_vs_oracle(ops, oracle_lib):
    """800x1344, batch 2, channels-last logits: per-level sums / gradients against the oracle
    (1e-5), with ignored anchors, every class id and logits beyond the exponential's clamp"""
    ph, pw, B = 800, 1344, 2
    geom, base = G.geometry(ph, pw, -1)
    cls, reg, iou = synth.head_outputs(31, B, ph, pw, 'A')
    cls = [x.copy() for x in cls]
    cls[1].reshape(-1)[::9973] = 75.0                              # > kXMax = 60
    rs = np.random.RandomState(5)
    labels, lw, bt, bw = [], [], [], []
    for (h, w) in geom.featmap_sizes:
        n = h * w * synth.A
        lab = np.zeros((B, n), np.int64)
        pos = rs.rand(B, n) < 0.004
        lab[pos] = rs.randint(1, 81, int(pos.sum()))
        wgt = (rs.rand(B, n) > 0.05).astype(np.float32)
        labels.append(lab); lw.append(wgt)
        bt.append((rs.standard_normal((B, n, 4)) * 0.2 * pos[..., None]).astype(np.float32))
        bw.append(np.repeat(pos[..., None].astype(np.float32), 4, -1))
    dev = lambda xs: [torch.from_numpy(x).cuda() for x in xs]    # noqa: E731
    c, r, i = _cl(G.to_dev(cls)), _cl(G.to_dev(reg)), _cl(G.to_dev(iou))
    avg = 37.0
    out = ops.head_loss(geom, c, r, i, dev(labels), dev(lw), dev(bt), dev(bw), avg_factor=avg,
                        exact_large_logits=True, channels_last=True)
    sum(v.total for v in out.values()).sum().backward()
    for l in range(geom.L):
        so, go = oracle_lib.focal_loss(cls[l], labels[l], lw[l], synth.A, 2.0, 0.25,
                                       gscale=1.0 / avg)
        assert rel(float(out['loss_cls'][l]), so / avg) < 1e-5, l
        g = c[l].grad.cpu().numpy()
        assert np.abs(g - go).max() <= 1e-5 * np.abs(go).max(), l
        s1, g1 = oracle_lib.smooth_l1(reg[l], bt[l], bw[l], synth.A, 0.11, gscale=1.0 / avg)
        s2, tgt, g_iou, g_box = oracle_lib.iou_bce(reg[l], iou[l], bt[l], bw[l], base[l],
                                                   synth.STRIDES[l], gscale=1.0 / avg)
        assert rel(float(out['loss_bbox'][l]), s1 / avg) < 1e-5, l
        assert rel(float(out['losses_iou'][l]), s2 / avg) < 1e-5, l
        gr = r[l].grad.cpu().numpy()
        assert np.abs(gr - (g1 + g_box)).max() <= 1e-6 * max(np.abs(g1 + g_box).max(), 1e-30), l
        gi = i[l].grad.cpu().numpy()
        assert np.abs(gi - g_iou).max() <= 1e-6 * max(np.abs(g_iou).max(), 1e-30), l


def test_anchor_targets_padded_entry_point_for_large_batches(ops):
    """batches beyond IA_MAX_TARGET_BATCH take ia_anchor_targets (padded gt tensor): same targets"""
    from iouaware import _lib
    B = _lib.IA_MAX_TARGET_BATCH + 1
    ph, pw = 128, 160
    geom, _ = G.geometry(ph, pw, -1)
    gts, gls = synth.train_targets(3, B, ph, pw, max_gt=6)
    gtb = [torch.from_numpy(x).cuda() for x in gts]
    gtl = [torch.from_numpy(x).cuda() for x in gls]
    big = ops.anchor_targets(geom, gtb, gtl, [(ph, pw, 3)] * B, 0.5, 0.4, 0.0, -1)
    for b0 in (0, 9):
        sl = slice(b0, b0 + 8)
        small = ops.anchor_targets(geom, gtb[sl], gtl[sl], [(ph, pw, 3)] * 8, 0.5, 0.4, 0.0, -1)
        assert torch.equal(big[4][sl], small[4])
        for k in range(4):
            for l in range(geom.L):
                assert torch.equal(big[k][l][sl], small[k][l])
