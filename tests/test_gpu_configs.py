"""GPU: every BASELINE.json configuration at ITS OWN SIZE in front of the oracle, collected at the front of the
`-m gpu` run (tests/conftest.py; VERDICT r5 item 2: "put every BASELINE config in front of the oracle early in
the order", done = `configs_untested: []`).

  config 2  R-50-FPN fp32, batch 8 at 800x1344: the post-conv path (reference iou_aware_retina_head.py:390-564 ->
            bbox_nms.py:6-67 -> nms_cpu.cpp:4-59) on a batch of 8, image 0 AND image 7 bit for bit against the
            oracle, in both head-output layouts (channels-last is what bench.py feeds it)
  config 3  R-101-FPN bf16, 16 images per GPU: the same on bf16 logits, image 0 and image 15
  config 4  X-101-64x4d at 800x1344: its reference fixture is tests/test_gpu_e2e.py
            `test_deeper_backbones_match_the_reference[x101_64x4d_full-*-winograd]` (bench path, tier 1)
  config 5  training step, 4 images per GPU at 800x1344 (configs/...r50_fpn_1x_4gpu.py:78): targets on the device
            (anchor_target.py:7-107, max_iou_assigner.py:50-201) equal the host restatement, then SigmoidFocalLoss +
            smooth-L1 + IoU-prediction BCE (iou_aware_retina_head.py:221-313, losses.py:226-303,385-480) forward sums
            and all three gradients against the oracle, both layouts
"""
import numpy as np
import pytest
import torch

import gpu_util as G
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available()
    from iouaware import ops as o
    return o


def _check_images(ops, oracle_lib, geom, base, host, dev, images, B):
    cls, reg, iou = host
    shapes, sfs = [(800, 1333, 3)] * B, [1.0] * B
    dets, labels, rows, num = [t.cpu().numpy() for t in
                               ops.get_bboxes(geom, *dev, shapes, sfs, True, 0.05, 0.5, 100)]
    for b in images:
        o = oracle_lib.get_bboxes_single([x[b] for x in cls], [x[b] for x in reg], [x[b] for x in iou],
                                         synth.STRIDES, base, (800, 1333), 1.0, True, 1000, 0.05, 0.5, 100)
        n = int(num[b])
        assert n == o['num_det'] and n > 0
        assert np.array_equal(rows[b, :n], o['det_rows']), b          # kept-box indices: bit-exact
        assert np.array_equal(labels[b, :n], o['det_labels']), b
        assert G.same_bits(dets[b, :n], o['det_bboxes']), b
    return dets, rows, num


@pytest.mark.parametrize('layout', ['channels_last', 'nchw'])
def test_config2_fp32_batch8_images_0_and_7_bit_exact(ops, oracle_lib, layout):
    ph, pw, B = 800, 1344, 8
    geom, base = G.geometry(ph, pw, 1000)
    host = synth.head_outputs(2024, B, ph, pw, 'C')
    dev = [G.to_dev(x) for x in host]
    if layout == 'channels_last':
        dev = [[t.contiguous(memory_format=torch.channels_last) for t in x] for x in dev]
        assert ops.geometry_for(geom, *dev).layout == 1
    _check_images(ops, oracle_lib, geom, base, host, dev, (0, 7), B)


def test_config2_dense_set_batch8_image_7_bit_exact(ops, oracle_lib):
    """SURVEY section 8(d) set B ("dense": every candidate passes score_thr, thousands of boxes per class into
    NMS) at batch 8, channels-last: the last image against the oracle"""
    ph, pw, B = 800, 1344, 8
    geom, base = G.geometry(ph, pw, 1000)
    host = synth.head_outputs(77, B, ph, pw, 'B')
    dev = [[t.contiguous(memory_format=torch.channels_last) for t in G.to_dev(x)] for x in host]
    _check_images(ops, oracle_lib, geom, base, host, dev, (7,), B)


def test_config3_bf16_batch16_post_conv_path(ops, oracle_lib):
    """BASELINE config 3's per-GPU shape: 16 images, bf16 head outputs at 800x1344, channels-last.  Image 0 and
    image 15 bit for bit against the oracle fed the same bf16-rounded logits; batch invariance for a middle image."""
    ph, pw, B = 800, 1344, 16
    geom, base = G.geometry(ph, pw, 1000)
    host = [G.bf16_round(x) for x in synth.head_outputs(777, B, ph, pw, 'C')]
    dev = [[t.contiguous(memory_format=torch.channels_last) for t in G.to_dev(x, torch.bfloat16)] for x in host]
    assert ops.geometry_for(geom, *dev).layout == 1
    dets, rows, num = _check_images(ops, oracle_lib, geom, base, host, dev, (0, 15), B)
    b = 7
    one = [t.cpu().numpy() for t in ops.get_bboxes(geom, *[[t[b:b + 1] for t in x] for x in dev],
                                                   [(800, 1333, 3)], [1.0], True, 0.05, 0.5, 100)]
    assert int(one[3][0]) == int(num[b]) and np.array_equal(one[0][0], dets[b])
    assert np.array_equal(one[2][0], rows[b])


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


@pytest.mark.parametrize('layout', ['channels_last', 'nchw'])
def test_config5_batch4_full_size_targets_and_losses_vs_oracle(ops, oracle_lib, layout):
    from iouaware.head import IoUawareRetinaHead
    from iouaware.targets import anchor_target
    from test_host_targets import HEAD_KW, TRAIN_CFG
    ph, pw, B = 800, 1344, 4
    head = IoUawareRetinaHead(**HEAD_KW)
    sizes = synth.level_shapes(ph, pw)
    geom = head.geometry(sizes, -1)
    base = oracle_lib.head_base_anchors(synth.STRIDES)
    gts, gls = synth.train_targets(55, B, 800, 1333, max_gt=20)                   # 1-20 gts per image
    metas = [synth.img_meta(800, 1333, ph, pw) for _ in range(B)]
    gtb = [torch.from_numpy(x).cuda() for x in gts]
    gtl = [torch.from_numpy(x).cuda() for x in gls]

    # ---- T1 / T2: targets made on the device == the host-side restatement of anchor_target (pinned on the
    # reference by tests/test_host_targets.py and test_gpu_targets.py's fixtures)
    labels, lw, bt, bw, counts = ops.anchor_targets(geom, gtb, gtl, [m['pad_shape'] for m in metas],
                                                    0.5, 0.4, 0.0, -1)
    anchors, flags = head.get_anchors(sizes, metas, device='cuda')
    ref = anchor_target(anchors, flags, gtb, metas, head.target_means, head.target_stds, TRAIN_CFG,
                        gt_labels_list=gtl, label_channels=80, sampling=False)
    num_total_pos = int(counts[:, 0].clamp(min=1).sum())
    assert num_total_pos == ref[4] and num_total_pos >= B
    for l in range(5):
        assert torch.equal(labels[l], ref[0][l].reshape(labels[l].shape))
        assert torch.equal(lw[l], ref[1][l].reshape(lw[l].shape))
        assert torch.equal(bw[l], ref[3][l].reshape(bw[l].shape))
        assert torch.allclose(bt[l], ref[2][l].reshape(bt[l].shape), rtol=1e-5, atol=1e-6)

    # ---- T3 / T3a / T3b / T3c: the all-levels loss node on those targets against the oracle
    cls, reg, iou = synth.head_outputs(56, B, ph, pw, 'A')
    c, r, i = G.to_dev(cls), G.to_dev(reg), G.to_dev(iou)
    cl = layout == 'channels_last'
    if cl:
        c, r, i = [[t.contiguous(memory_format=torch.channels_last) for t in x] for x in (c, r, i)]
    c, r, i = [[t.requires_grad_(True) for t in x] for x in (c, r, i)]
    avg = float(num_total_pos)
    out = ops.head_loss(geom, c, r, i, labels, lw, bt, bw, avg_factor=avg, channels_last=cl)
    sum(v.total for v in out.values()).sum().backward()
    torch.cuda.synchronize()
    for l in range(geom.L):
        lab, wgt = labels[l].cpu().numpy(), lw[l].cpu().numpy()
        tgt, twt = bt[l].cpu().numpy(), bw[l].cpu().numpy()
        so, go = oracle_lib.focal_loss(cls[l], lab, wgt, synth.A, 2.0, 0.25, gscale=1.0 / avg)
        assert _rel(float(out['loss_cls'][l]), so / avg) < 1e-5, l
        g = c[l].grad.cpu().numpy()
        assert np.abs(g - go).max() <= 1e-5 * np.abs(go).max(), l
        s1, g1 = oracle_lib.smooth_l1(reg[l], tgt, twt, synth.A, 0.11, gscale=1.0 / avg)
        s2, _, g_iou, g_box = oracle_lib.iou_bce(reg[l], iou[l], tgt, twt, base[l], synth.STRIDES[l],
                                                 gscale=1.0 / avg)
        assert _rel(float(out['loss_bbox'][l]), s1 / avg) < 1e-5, l
        assert _rel(float(out['losses_iou'][l]), s2 / avg) < 1e-5, l
        gr = r[l].grad.cpu().numpy()
        assert np.abs(gr - (g1 + g_box)).max() <= 1e-6 * max(np.abs(g1 + g_box).max(), 1e-30), l
        gi = i[l].grad.cpu().numpy()
        assert np.abs(gi - g_iou).max() <= 1e-6 * max(np.abs(g_iou).max(), 1e-30), l
