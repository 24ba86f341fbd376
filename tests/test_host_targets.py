"""CPU: training-target assignment against the targets the
reference's anchor_target produced for the same gt boxes (tests/golden/losses_small.npz)."""
import os

import numpy as np
import torch

import synth
from iouaware.config import ConfigDict
from iouaware.head import IoUawareRetinaHead
from iouaware.targets import anchor_target, MaxIoUAssigner
import pytest

HEAD_KW = dict(num_classes=81, in_channels=256, stacked_convs=4, feat_channels=256,
               octave_base_scale=4, scales_per_octave=3, anchor_ratios=[0.5, 1.0, 2.0],
               anchor_strides=[8, 16, 32, 64, 128], target_means=[.0] * 4, target_stds=[1.0] * 4,
               loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25,
                             loss_weight=1.0),
               loss_bbox=dict(type='SmoothL1Loss', beta=0.11, loss_weight=1.0))
TRAIN_CFG = ConfigDict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4,
                                     min_pos_iou=0, ignore_iof_thr=-1),
                       allowed_border=-1, pos_weight=-1, debug=False)


def test_anchor_target_equals_reference(golden_dir):
    f = np.load(os.path.join(golden_dir, 'losses_small.npz'))
    ih, iw, ph, pw = [int(v) for v in f['img']]
    B = int(f['batch'])
    head = IoUawareRetinaHead(**HEAD_KW)
    sizes = synth.level_shapes(ph, pw)
    metas = [synth.img_meta(ih, iw, ph, pw) for _ in range(B)]
    anchors, flags = head.get_anchors(sizes, metas)
    gts = [torch.from_numpy(f['gt_bboxes_%d' % b]) for b in range(B)]
    gls = [torch.from_numpy(f['gt_labels_%d' % b]) for b in range(B)]
    out = anchor_target(anchors, flags, gts, metas, head.target_means, head.target_stds, TRAIN_CFG,
                        gt_labels_list=gls, label_channels=80, sampling=False)
    labels, lw, bt, bw, npos, nneg, lvl_anchors = out
    assert npos == int(f['num_total_pos']) and nneg == int(f['num_total_neg'])
    for l in range(5):
        assert np.array_equal(labels[l].numpy(), f['labels_%d' % l])
        assert np.array_equal(lw[l].numpy(), f['label_weights_%d' % l])
        assert np.array_equal(bw[l].numpy(), f['bbox_weights_%d' % l])
        assert np.allclose(bt[l].numpy(), f['bbox_targets_%d' % l], rtol=1e-6, atol=1e-6)


def test_assigner_rules():
    a = MaxIoUAssigner(0.5, 0.4, min_pos_iou=0.0)
    boxes = torch.tensor([[0., 0., 9., 9.], [0., 0., 4., 9.], [50., 50., 59., 59.],
                          [100., 100., 101., 101.]])
    gt = torch.tensor([[0., 0., 9., 9.], [52., 52., 61., 61.]])
    r = a.assign(boxes, gt, gt_labels=torch.tensor([3, 7]))
    # box0 IoU 1 -> gt1; box1 IoU .5 -> positive (>=); box2 IoU ~.47 -> claimed as gt2's best;
    # box3 far -> negative
    assert r.gt_inds.tolist() == [1, 1, 2, 0]
    assert r.labels.tolist() == [3, 3, 7, 0]
    with pytest.raises(ValueError, match='No gt or bboxes'):
        a.assign(boxes, torch.zeros(0, 4))
