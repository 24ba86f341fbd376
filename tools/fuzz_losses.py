#!/usr/bin/env python
"""Randomised check of the loss kernels against the fp64-summing C oracle:
  * ops.head_loss (all-levels focal + smooth-L1 + IoU-BCE, forward sums and every gradient) on random
    pyramid sizes, batches, positive rates (incl. no positive at all), ignored anchors, NCHW and
    channels-last heads;
  * mmdet.ops.sigmoid_focal_loss (iouaware.focal_op) on random (N, C), targets incl. -1 / background,
    gamma / alpha -- bit for bit, forward and backward.
    python tools/fuzz_losses.py [cases] [first seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tests', 'oracle', 'iou-aware-single-stage-object-detector_amd', ''):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gpu_util as G  # noqa: E402
import synth  # noqa: E402
import oracle  # noqa: E402
from iouaware import ops  # noqa: E402
from iouaware.focal_op import sigmoid_focal_loss  # noqa: E402

oracle.build()


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


def head_case(seed):
    rs = np.random.RandomState(seed)
    ph, pw = 32 * int(rs.randint(2, 14)), 32 * int(rs.randint(2, 18))
    B = int(rs.randint(1, 5))
    rate = float(rs.choice([0.0, 0.0005, 0.004, 0.05]))
    cl = bool(rs.rand() < 0.5)
    kind = str(rs.choice(['A', 'B']))
    avg = float(rs.choice([1.0, 37.0, 1234.0]))
    tag = 'seed %d head_loss %dx%d B=%d positives=%.4f channels_last=%d kind=%s avg=%g' % (seed, ph, pw, B, rate, cl, kind, avg)
    geom, base = G.geometry(ph, pw, -1)
    cls, reg, iou = synth.head_outputs(seed, B, ph, pw, kind)
    labels, lw, bt, bw = [], [], [], []
    for (h, w) in geom.featmap_sizes:
        n = h * w * synth.A
        lab = np.zeros((B, n), np.int64)
        pos = rs.rand(B, n) < rate
        lab[pos] = rs.randint(1, 81, int(pos.sum()))
        labels.append(lab)
        lw.append((rs.rand(B, n) > 0.05).astype(np.float32))
        bt.append((rs.standard_normal((B, n, 4)) * 0.2 * pos[..., None]).astype(np.float32))
        bw.append(np.repeat(pos[..., None].astype(np.float32), 4, -1))
    dev = lambda xs: [torch.from_numpy(x).cuda() for x in xs]    # noqa: E731

    def leaf(ts):
        ts = G.to_dev(ts)
        if cl:
            ts = [t.contiguous(memory_format=torch.channels_last) for t in ts]
        return [t.requires_grad_(True) for t in ts]
    c, r, i = leaf(cls), leaf(reg), leaf(iou)
    out = ops.head_loss(geom, c, r, i, dev(labels), dev(lw), dev(bt), dev(bw), avg_factor=avg)
    sum(v.total for v in out.values()).sum().backward()
    for l in range(geom.L):
        so, go = oracle.focal_loss(cls[l], labels[l], lw[l], synth.A, 2.0, 0.25, gscale=1.0 / avg)
        assert rel(float(out['loss_cls'][l]), so / avg) < 1e-5, tag + ' focal sum level %d' % l
        assert np.abs(c[l].grad.cpu().numpy() - go).max() <= 1e-5 * max(np.abs(go).max(), 1e-30), tag + ' focal grad level %d' % l
        s1, g1 = oracle.smooth_l1(reg[l], bt[l], bw[l], synth.A, 0.11, gscale=1.0 / avg)
        s2, tgt, g_iou, g_box = oracle.iou_bce(reg[l], iou[l], bt[l], bw[l], base[l], synth.STRIDES[l], gscale=1.0 / avg)
        assert abs(float(out['loss_bbox'][l]) - s1 / avg) <= 1e-5 * abs(s1 / avg) + 1e-12, tag + ' smooth-l1 sum level %d' % l
        assert abs(float(out['losses_iou'][l]) - s2 / avg) <= 1e-5 * abs(s2 / avg) + 1e-12, tag + ' iou sum level %d' % l
        want = g1 + g_box
        assert np.abs(r[l].grad.cpu().numpy() - want).max() <= 1e-6 * max(np.abs(want).max(), 1e-30) + 1e-12, tag + ' reg grad level %d' % l
        assert np.abs(i[l].grad.cpu().numpy() - g_iou).max() <= 1e-6 * max(np.abs(g_iou).max(), 1e-30) + 1e-12, tag + ' iou grad level %d' % l
    return tag


def op_case(seed):
    rs = np.random.RandomState(seed)
    N, Cn = int(rs.choice([1, 7, 300, 5000])), int(rs.choice([1, 2, 80, 81]))
    gamma, alpha = float(rs.choice([2.0, 1.5, 0.0])), float(rs.choice([0.25, 0.5]))
    tag = 'seed %d focal op N=%d C=%d gamma=%g alpha=%g' % (seed, N, Cn, gamma, alpha)
    x = (rs.standard_normal((N, Cn)) * float(rs.choice([1, 4, 30]))).astype(np.float32)
    t = rs.randint(-1, Cn + 1, N).astype(np.int64)
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    out = sigmoid_focal_loss(xd, torch.from_numpy(t).cuda(), gamma, alpha, 'none')
    assert G.same_bits(out.detach().cpu().numpy(), oracle.focal_loss_op(x, t, gamma, alpha)), tag + ' forward'
    dl = rs.uniform(0.5, 1.5, (N, Cn)).astype(np.float32)
    out.backward(torch.from_numpy(dl).cuda())
    assert G.same_bits(xd.grad.cpu().numpy(), oracle.focal_loss_op(x, t, gamma, alpha, dl)), tag + ' backward'
    return tag


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
    bad, t0 = 0, time.time()
    for i in range(cases):
        try:
            print('ok   ' + (head_case if i % 2 == 0 else op_case)(seed0 + i), flush=True)
        except Exception as exc:
            bad += 1
            print('FAIL seed %d -> %s: %s' % (seed0 + i, type(exc).__name__, str(exc)[:400]), flush=True)
    print('%d cases, %d failures, %.0f s' % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
