#!/usr/bin/env python
"""Randomised training-iteration check (BASELINE config 5's path): R-50 IoU-aware RetinaNet, trained-like
weights, one forward_train -> parse_losses -> backward through the FUSED training route (own autograd
nodes: GEMM / Winograd convolutions, folded eval-mode BatchNorm, device targets, all-levels loss node)
against the plain torch modules -- random pad sizes (down to 2 x 2 pyramid maps), batches, image shapes
inside the pad, 1 ... 40 gt boxes incl. tiny ones.  Loss within 1e-4, every parameter gradient's
norm-wise deviation < 5e-2, all gradients together < 5e-3 (tests/test_gpu_train_fuse.py's bounds).
    python tools/fuzz_train.py [cases] [first seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tests', 'oracle', 'iou-aware-single-stage-object-detector_amd', ''):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
import iouaware  # noqa: E402
import bench  # noqa: E402
from iouaware.config import ConfigDict  # noqa: E402
from iouaware.fuse import fuse_inference  # noqa: E402
from iouaware.train import parse_losses  # noqa: E402


def rel2(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def run_case(seed):
    rs = np.random.RandomState(seed)
    ph, pw = 32 * int(rs.randint(2, 11)), 32 * int(rs.randint(2, 13))
    B = int(rs.randint(1, 4))
    ih, iw = ph - int(rs.randint(0, 32)), pw - int(rs.randint(0, 32))
    max_gt = int(rs.choice([1, 3, 12, 40]))
    gts, gls = synth.train_targets(seed, B, ih, iw, max_gt=max_gt)
    if rs.rand() < 0.3:                                    # a tiny box: positive only through the low-quality match
        gts[0][0] = np.asarray([5, 7, 9, 12], np.float32)
    tag = 'seed %d: %dx%d (img %dx%d) B=%d gts=%s' % (seed, ph, pw, ih, iw, B, [len(g) for g in gts])
    gtb = [torch.from_numpy(x).cuda() for x in gts]
    gtl = [torch.from_numpy(x).cuda() for x in gls]
    metas = [synth.img_meta(ih, iw, ph, pw) for _ in range(B)]
    img = torch.from_numpy(synth.e2e_image(seed, B, ph, pw, ih, iw)).cuda()
    res = {}
    for mode in ('ref', 'fused'):
        torch.manual_seed(0)
        model = iouaware.build_detector(ConfigDict(bench.MODEL), train_cfg=ConfigDict(bench.TRAIN_CFG),
                                        test_cfg=ConfigDict(bench.TEST_CFG))
        state = model.state_dict()
        synth.e2e_fill_state(state, 7)
        model.load_state_dict(state)
        model = model.cuda().train()
        x = img
        if mode == 'fused':
            model.bbox_head.train_winograd = True
            assert fuse_inference(model, winograd=True, train=True) > 0
            x = img.contiguous(memory_format=torch.channels_last)
        else:
            model.bbox_head.train_winograd = False
        losses = model(x, metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
        loss, _ = parse_losses(losses)
        loss.backward()
        res[mode] = (float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    (la, ga), (lb, gb) = res['fused'], res['ref']
    assert np.isfinite(la) and abs(la - lb) <= 1e-4 * abs(lb), tag + ' loss %r vs %r' % (la, lb)
    assert set(ga) == set(gb), tag + ' gradient sets differ'
    worst = max((rel2(ga[k], gb[k]), k) for k in gb)
    assert worst[0] < 5e-2, tag + ' %s' % (worst,)
    tot = rel2(torch.cat([g.flatten() for g in ga.values()]), torch.cat([gb[k].flatten() for k in ga]))
    assert tot < 5e-3, tag + ' all gradients %.2e' % tot
    return tag + '  (loss %.4f, worst gradient %.1e, all %.1e)' % (la, worst[0], tot)


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    bad, t0 = 0, time.time()
    for i in range(cases):
        try:
            print('ok   ' + run_case(seed0 + i), flush=True)
        except Exception as exc:
            bad += 1
            print('FAIL seed %d -> %s: %s' % (seed0 + i, type(exc).__name__, str(exc)[:400]), flush=True)
    print('%d cases, %d failures, %.0f s' % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
