"""CPU: the oracle's training losses (oracle/iouaware_oracle_loss.c) against the
reference's own loss values and autograd gradients captured in
tests/golden/losses_small.npz (head.loss on a 2-image batch, per level).

Bar: loss sums and gradients within 1e-4 relative (sums: of the value;
gradients: of the tensor's max magnitude -- elementwise roundoff of the
reference's own fp32 autograd chain is of that order)."""
import os

import numpy as np
import pytest

import synth


@pytest.fixture(scope='module')
def fx(golden_dir):
    f = np.load(os.path.join(golden_dir, 'losses_small.npz'))
    ih, iw, ph, pw = [int(v) for v in f['img']]
    B = int(f['batch'])
    cls, reg, iou = synth.head_outputs(int(f['seed']), B, ph, pw, str(f['kind']))
    assert synth.checksum(cls + reg + iou) == int(f['checksum'])
    return f, cls, reg, iou, B, synth.level_shapes(ph, pw)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def grad_close(g, f, key, mode, tol=1e-4):
    idx = f[key + '_idx']
    want = f['%s_%s' % (key, mode)].astype(np.float64)
    got = g.reshape(-1)[idx].astype(np.float64)
    scale = max(np.abs(want).max(), 1e-30)
    assert np.abs(got - want).max() <= tol * scale, (key, np.abs(got - want).max(), scale)
    assert rel(np.abs(g.astype(np.float64)).sum(), float(f['%s_%s_abs' % (key, mode)])) < 1e-4


def test_losses_and_grads_match_reference(oracle_lib, fx):
    f, cls, reg, iou, B, shapes = fx
    base = oracle_lib.head_base_anchors(synth.STRIDES)
    avg = float(f['num_total_pos'])
    for l, (h, w) in enumerate(shapes):
        labels, lw = f['labels_%d' % l].reshape(-1), f['label_weights_%d' % l].reshape(-1)
        bt, bw = f['bbox_targets_%d' % l].reshape(-1, 4), f['bbox_weights_%d' % l].reshape(-1, 4)
        s, g = oracle_lib.focal_loss(cls[l], labels, lw, synth.A, 2.0, 0.25, gscale=1.0 / avg)
        assert rel(s / avg, f['loss_cls'][l]) < 1e-4
        grad_close(g, f, 'g_cls_%d' % l, 'attached')
        s, g = oracle_lib.smooth_l1(reg[l], bt, bw, synth.A, 0.11, gscale=1.0 / avg)
        assert abs(s / avg - f['loss_bbox'][l]) <= 1e-4 * max(f['loss_bbox'][l], 1e-6)
        s2, tgt, g_iou, g_box = oracle_lib.iou_bce(reg[l], iou[l], bt, bw, base[l],
                                                   synth.STRIDES[l], gscale=1.0 / avg)
        assert abs(s2 / avg - f['losses_iou'][l]) <= 1e-4 * max(f['losses_iou'][l], 1e-6)
        assert ((tgt >= 0) & (tgt <= 1.0 + 1e-6)).all()
        grad_close(g_iou, f, 'g_iou_%d' % l, 'attached')
        # bbox_pred receives smooth-L1 grad (+ the IoU-target path when attached)
        grad_close(g, f, 'g_reg_%d' % l, 'detached', tol=2e-4)
        grad_close(g + g_box, f, 'g_reg_%d' % l, 'attached', tol=2e-4)


def test_focal_op_formula_against_float64(oracle_lib):
    """T4: the reference CUDA op has no CPU implementation (sigmoid_focal_loss.cpp:21-25 falls
    through), so its formula (sigmoid_focal_loss_cuda.cu:23-105) is pinned against float64."""
    rs = np.random.RandomState(4)
    N, Cn, gamma, alpha = 500, 80, 2.0, 0.25
    x = (rs.standard_normal((N, Cn)) * 4).astype(np.float32)
    t = rs.randint(-1, Cn + 1, N).astype(np.int64)        # -1 = ignored rows
    xd = x.astype(np.float64)
    p = 1 / (1 + np.exp(-xd))
    d = np.arange(Cn)[None, :]
    c1 = (t[:, None] == d + 1).astype(np.float64)
    c2 = ((t[:, None] >= 0) & (t[:, None] != d + 1)).astype(np.float64)
    lg2 = -xd * (xd >= 0) - np.log1p(np.exp(xd - 2 * xd * (xd >= 0)))
    want = -c1 * (1 - p) ** gamma * np.log(np.maximum(p, 1.17549435e-38)) * alpha \
        - c2 * p ** gamma * lg2 * (1 - alpha)
    got = oracle_lib.focal_loss_op(x, t, gamma, alpha)
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    dl = rs.uniform(0.5, 1.5, (N, Cn)).astype(np.float32)
    gwant = (-c1 * (1 - p) ** gamma * (1 - p - p * gamma * np.log(np.maximum(p, 1.17549435e-38)))
             * alpha - c2 * p ** gamma * (lg2 * (1 - p) * gamma - p) * (1 - alpha)) * dl
    ggot = oracle_lib.focal_loss_op(x, t, gamma, alpha, dl)
    assert np.abs(ggot - gwant).max() <= 1e-5 * max(1.0, np.abs(gwant).max())


def test_focal_op_pinned_on_reference_py_focal_loss(oracle_lib, golden_dir):
    """T4 pin: for integer targets and unit weights the reference's own CPU
    py_sigmoid_focal_loss (losses.py:226-247) on the one-hot target computes what the CUDA op
    computes; loss and autograd gradient captured by make_golden.py focal_op."""
    f = np.load(os.path.join(golden_dir, 'focal_op.npz'))
    x, t, up = f['logits'], f['targets'], f['upstream']
    for gamma, alpha in f['params']:
        tag = 'g%g_a%g' % (gamma, alpha)
        got = oracle_lib.focal_loss_op(x, t, float(gamma), float(alpha))
        want = f['loss_' + tag]
        assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), tag
        assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), tag
        ggot = oracle_lib.focal_loss_op(x, t, float(gamma), float(alpha), up)
        gwant = f['grad_' + tag]
        assert np.abs(ggot - gwant).max() <= 1e-5 * max(1.0, np.abs(gwant).max()), tag


def test_iou_balanced_losses_match_reference(oracle_lib, fx, golden_dir):
    """IOUbalancedSigmoidFocalLoss(eta=1.5) + IoUbalancedSmoothL1Loss(delta=1.5,
    loss_weight=3.049) through the reference head (tests/golden/losses_balanced.npz); targets are
    those of losses_small (same inputs)."""
    f, cls, reg, iou, B, shapes = fx
    fb = np.load(os.path.join(golden_dir, 'losses_balanced.npz'))
    assert int(fb['checksum']) == int(f['checksum'])
    base = oracle_lib.head_base_anchors(synth.STRIDES)
    avg = float(f['num_total_pos'])
    eta, delta, lwt = float(fb['eta']), float(fb['delta']), float(fb['bbox_loss_weight'])

    def gclose(g, key, tol=2e-4):
        want = fb[key].astype(np.float64)
        got = g.reshape(-1)[fb[key + '_idx']].astype(np.float64)
        scale = max(np.abs(want).max(), 1e-30)
        assert np.abs(got - want).max() <= tol * scale, (key, np.abs(got - want).max(), scale)
        assert rel(np.abs(g.astype(np.float64)).sum(), float(fb[key + '_abs'])) < 2e-4

    for l, (h, w) in enumerate(shapes):
        labels, lw = f['labels_%d' % l].reshape(-1), f['label_weights_%d' % l].reshape(-1)
        bt, bw = f['bbox_targets_%d' % l].reshape(-1, 4), f['bbox_weights_%d' % l].reshape(-1, 4)
        s2, tgt, g_iou, g_box = oracle_lib.iou_bce(reg[l], iou[l], bt, bw, base[l],
                                                   synth.STRIDES[l], gscale=1.0 / avg)
        s, g, sums = oracle_lib.focal_loss_balanced(cls[l], labels, lw, tgt, synth.A, 2.0, 0.25,
                                                    eta, gscale=1.0 / avg)
        assert rel(s / avg, fb['loss_cls'][l]) < 1e-4
        gclose(g, 'g_cls_%d' % l)
        sb, gb = oracle_lib.smooth_l1_balanced(reg[l], bt, bw, tgt, synth.A, 0.11, delta,
                                               gscale=lwt / avg)
        assert abs(sb * lwt / avg - fb['loss_bbox'][l]) <= 1e-4 * max(fb['loss_bbox'][l], 1e-6)
        assert abs(s2 / avg - fb['losses_iou'][l]) <= 1e-4 * max(fb['losses_iou'][l], 1e-6)
        gclose(gb + g_box, 'g_reg_%d' % l)
        gclose(g_iou, 'g_iou_%d' % l)
