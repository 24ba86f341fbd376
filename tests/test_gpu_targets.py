"""GPU: device target assignment (csrc/assign.hip) against the targets the
reference's anchor_target produced (tests/golden/losses_small.npz) and against the torch
implementation in iouaware/targets.py on random batches."""
import os

import numpy as np
import pytest
import torch

import synth
import gpu_util as G

pytestmark = pytest.mark.gpu


def test_device_targets_equal_reference(golden_dir):
    from iouaware import ops
    f = np.load(os.path.join(golden_dir, 'losses_small.npz'))
    ih, iw, ph, pw = [int(v) for v in f['img']]
    B = int(f['batch'])
    geom, base = G.geometry(ph, pw, -1)
    gts = [torch.from_numpy(f['gt_bboxes_%d' % b]).cuda() for b in range(B)]
    gls = [torch.from_numpy(f['gt_labels_%d' % b]).cuda() for b in range(B)]
    labels, lw, bt, bw, counts = ops.anchor_targets(geom, gts, gls, [(ph, pw, 3)] * B, 0.5, 0.4, 0.0,
                                                    -1)
    assert int(counts[:, 0].clamp(min=1).sum()) == int(f['num_total_pos'])
    assert int(counts[:, 1].clamp(min=1).sum()) == int(f['num_total_neg'])
    for l in range(5):
        assert np.array_equal(labels[l].cpu().numpy(), f['labels_%d' % l])
        assert np.array_equal(lw[l].cpu().numpy(), f['label_weights_%d' % l])
        assert np.array_equal(bw[l].cpu().numpy(), f['bbox_weights_%d' % l])
        assert np.allclose(bt[l].cpu().numpy(), f['bbox_targets_%d' % l], rtol=1e-5, atol=1e-6)


def test_device_targets_and_losses_mixed_pad_shapes_equal_reference(golden_dir):
    """T1 pinned on the reference: images of different pad_shape in one batch (partly false
    valid_flags / inside_flags, `unmap`): ia_anchor_targets and the whole head.loss (device
    targets + all-levels loss kernels, both layouts) against tests/golden/losses_mixed_pad.npz"""
    from iouaware import ops
    from iouaware.head import IoUawareRetinaHead
    from test_host_targets import HEAD_KW, TRAIN_CFG
    f = np.load(os.path.join(golden_dir, 'losses_mixed_pad.npz'))
    ph, pw = [int(v) for v in f['tensor']]
    B = int(f['batch'])
    shapes = [[int(v) for v in f['shapes'][b]] for b in range(B)]
    geom, base = G.geometry(ph, pw, -1)
    gts = [torch.from_numpy(f['gt_bboxes_%d' % b]).cuda() for b in range(B)]
    gls = [torch.from_numpy(f['gt_labels_%d' % b]).cuda() for b in range(B)]
    pads = [(s[2], s[3], 3) for s in shapes]
    labels, lw, bt, bw, counts = ops.anchor_targets(geom, gts, gls, pads, 0.5, 0.4, 0.0, -1)
    assert int(counts[:, 0].clamp(min=1).sum()) == int(f['num_total_pos'])
    assert int(counts[:, 1].clamp(min=1).sum()) == int(f['num_total_neg'])
    for l in range(5):
        assert np.array_equal(labels[l].cpu().numpy(), f['labels_%d' % l])
        assert np.array_equal(lw[l].cpu().numpy(), f['label_weights_%d' % l])
        assert np.array_equal(bw[l].cpu().numpy(), f['bbox_weights_%d' % l])
        assert np.allclose(bt[l].cpu().numpy(), f['bbox_targets_%d' % l], rtol=1e-5, atol=1e-6)
        inval = torch.from_numpy(f['valid_1_%d' % l] == 0).cuda()
        assert int(inval.sum()) > 0 and float(lw[l][1][inval].abs().sum()) == 0.0
    # the loss dict and the gradients of the head outputs, as the reference's autograd gives them
    cls, reg, iou = synth.head_outputs(int(f['seed']), B, ph, pw, str(f['kind']))
    assert synth.checksum(cls + reg + iou) == int(f['checksum'])
    head = IoUawareRetinaHead(**HEAD_KW).cuda()
    metas = [synth.img_meta(*s) for s in shapes]
    for channels_last in (False, True):
        c, r, i = [[t.contiguous(memory_format=torch.channels_last) if channels_last else t
                    for t in G.to_dev(x)] for x in (cls, reg, iou)]
        for t in c + r + i:
            t.requires_grad_(True)
        losses = head.loss(c, r, i, gts, gls, metas, TRAIN_CFG)
        for k in ('loss_cls', 'loss_bbox', 'losses_iou'):
            got = np.array([float(x) for x in losses[k]])
            assert np.all(np.abs(got - f[k]) <= 1e-4 * np.maximum(np.abs(f[k]), 1e-6)), (k, got, f[k])
        sum(sum(v) for v in losses.values()).backward()
        for l in range(5):
            for key, g in (('g_cls_%d' % l, c[l].grad), ('g_reg_%d' % l, r[l].grad),
                           ('g_iou_%d' % l, i[l].grad)):
                want = f[key].astype(np.float64)
                got = g.contiguous().cpu().numpy().reshape(-1)[f[key + '_idx']].astype(np.float64)
                assert np.abs(got - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-30), key
                tot = float(g.double().sum())
                assert abs(tot - float(f[key + '_sum'])) <= 2e-4 * max(float(f[key + '_abs']), 1e-30)


@pytest.mark.parametrize('seed,pad', [(1, (800, 1344)), (2, (320, 416)), (3, (608, 1024))])
def test_device_targets_equal_torch_path(seed, pad):
    """full-size and odd-size batches, padded images (valid flags), many gts"""
    from iouaware import ops
    from iouaware.head import IoUawareRetinaHead
    from iouaware.targets import anchor_target
    from test_host_targets import HEAD_KW, TRAIN_CFG
    ph, pw = pad
    B = 3
    head = IoUawareRetinaHead(**HEAD_KW)
    sizes = synth.level_shapes(ph, pw)
    geom = head.geometry(sizes, -1)
    gts, gls = synth.train_targets(seed, B, ph - 40, pw - 70, max_gt=30)
    # image 0 narrower than the pad: part of the feature map is invalid
    metas = [synth.img_meta(ph - 40, pw - 70, ph - (32 if b == 0 else 0), pw - (64 if b == 0 else 0))
             for b in range(B)]
    gtb = [torch.from_numpy(x).cuda() for x in gts]
    gtl = [torch.from_numpy(x).cuda() for x in gls]
    anchors, flags = head.get_anchors(sizes, metas, device='cuda')
    ref = anchor_target(anchors, flags, gtb, metas, head.target_means, head.target_stds, TRAIN_CFG,
                        gt_labels_list=gtl, label_channels=80, sampling=False)
    labels, lw, bt, bw, counts = ops.anchor_targets(geom, gtb, gtl, [m['pad_shape'] for m in metas],
                                                    0.5, 0.4, 0.0, -1)
    assert int(counts[:, 0].clamp(min=1).sum()) == ref[4]
    assert int(counts[:, 1].clamp(min=1).sum()) == ref[5]
    for l in range(5):
        assert torch.equal(labels[l], ref[0][l].reshape(labels[l].shape))
        assert torch.equal(lw[l], ref[1][l].reshape(lw[l].shape))
        assert torch.equal(bw[l], ref[3][l].reshape(bw[l].shape))
        assert torch.allclose(bt[l], ref[2][l].reshape(bt[l].shape), rtol=1e-5, atol=1e-6)
