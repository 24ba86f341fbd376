"""GPU: training through the head on the Winograd path (iouaware/winograd_train.py) against the
plain nn.Module head (MIOpen convolutions + autograd): outputs and every parameter / input
gradient within 1e-4 of their scale; a whole training iteration gives the same losses."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _head():
    from iouaware.head import IoUawareRetinaHead
    from test_host_targets import HEAD_KW
    torch.manual_seed(3)
    head = IoUawareRetinaHead(**HEAD_KW).cuda().train()
    with torch.no_grad():                      # activations of unit scale through the towers
        for p in head.parameters():
            if p.dim() == 4:
                p.normal_(0, (2.0 / (9 * p.shape[1])) ** 0.5)
            else:
                p.normal_(0, 0.1)
    return head


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def _rel2(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


@pytest.mark.parametrize('cin,cout,bias', [(256, 256, True), (256, 720, True), (256, 48, False),
                                           (64, 36, True)])
def test_conv_levels_node_vs_conv2d_autograd(cin, cout, bias):
    """one shared-weight convolution over three levels, no ReLU (no mask flips): output, input
    gradient and weight / bias gradients against F.conv2d's autograd, 1e-4 of their scale"""
    import torch.nn.functional as F
    from iouaware.winograd_train import wino_conv_levels
    g = torch.Generator(device='cuda').manual_seed(cin + cout)
    w = (torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * (1.0 / (9 * cin)) ** 0.5)
    b = torch.randn(cout, device='cuda', generator=g) * 0.1 if bias else None
    shapes = [(2, cin, 37, 53), (2, cin, 19, 27), (2, cin, 5, 3)]
    xs0 = [torch.randn(s, device='cuda', generator=g) for s in shapes]
    ups = [torch.randn((s[0], cout, s[2], s[3]), device='cuda', generator=g) for s in shapes]
    out = {}
    for mode in ('wino', 'ref'):
        wp = w.clone().requires_grad_(True)
        bp = b.clone().requires_grad_(True) if bias else None
        xs = [x.clone().requires_grad_(True) for x in xs0]
        ys = wino_conv_levels(xs, wp, bp) if mode == 'wino' else [F.conv2d(x, wp, bp, padding=1)
                                                                  for x in xs]
        sum((y * u).sum() for y, u in zip(ys, ups)).backward()
        out[mode] = ([y.detach() for y in ys], [x.grad for x in xs], wp.grad,
                     bp.grad if bias else None)
    (ya, xa, wa, ba), (yb, xb, wb, bb) = out['wino'], out['ref']
    for a, c in zip(ya + xa, yb + xb):
        assert a.shape == c.shape and _rel(a, c) < 1e-4
    assert _rel(wa, wb) < 1e-4
    if bias:
        assert _rel(ba, bb) < 1e-4


@pytest.mark.parametrize('layout', ['nchw', 'channels_last', 'all_active'])
def test_head_forward_backward_matches_module_path(layout):
    head = _head()
    strict = layout == 'all_active'
    if strict:
        # tower biases large enough that no pre-activation is ever negative: every ReLU is the
        # identity in both paths, the head is linear, and gradients must agree element-wise
        with torch.no_grad():
            for m in list(head.cls_convs) + list(head.reg_convs):
                m.conv.weight.mul_(0.2)
                m.conv.bias.fill_(6.0)
    g = torch.Generator(device='cuda').manual_seed(1)
    sizes = synth.level_shapes(224, 288)
    feats = [torch.randn(2, 256, h, w, device='cuda', generator=g) for (h, w) in sizes]
    if layout == 'channels_last':
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    ups = None
    res = {}
    for mode in (True, False):
        head.train_winograd = mode
        head.zero_grad()
        xs = [f.clone().requires_grad_(True) for f in feats]
        cls, reg, iou = head(xs)
        if ups is None:
            ups = [[torch.randn(t.shape, device='cuda', generator=g) for t in o]
                   for o in (cls, reg, iou)]
        loss = sum((t * u).sum() for o, us in zip((cls, reg, iou), ups) for t, u in zip(o, us))
        loss.backward()
        res[mode] = ([t.detach().contiguous() for o in (cls, reg, iou) for t in o],
                     {n: p.grad.clone() for n, p in head.named_parameters()},
                     [x.grad.contiguous() for x in xs])
    (oa, ga, xa), (ob, gb, xb) = res[True], res[False]
    assert [tuple(t.shape) for t in oa] == [tuple(t.shape) for t in ob]
    for a, b in zip(oa, ob):
        assert _rel(a, b) < 1e-4
    # Gradients pass through four ReLUs per tower: a pre-activation within 1e-5 of zero may fall
    # on the other side of the mask in the two paths, which moves single gradient elements by
    # O(1) under this test's random upstream gradients.  Hence a norm-wise bound (and a loose
    # element-wise one); the mask-free node is checked element-wise above.
    tol2, tol = (1e-4, 2e-4) if strict else (5e-2, 1.0)
    for n in gb:
        assert ga[n].shape == gb[n].shape and _rel2(ga[n], gb[n]) < tol2, (n, _rel2(ga[n], gb[n]))
        assert _rel(ga[n], gb[n]) < tol, n
    for a, b in zip(xa, xb):
        assert _rel2(a, b) < tol2


def test_training_iteration_same_losses_and_grads():
    """whole detector, one iteration: losses and head / FPN gradients with and without the
    Winograd training path"""
    import bench
    import iouaware
    from iouaware.config import ConfigDict
    from iouaware.train import parse_losses
    from test_host_targets import TRAIN_CFG
    torch.manual_seed(0)
    model = iouaware.build_detector(ConfigDict(bench.MODEL), train_cfg=TRAIN_CFG,
                                    test_cfg=ConfigDict(bench.TEST_CFG)).cuda().train()
    B, ph, pw = 2, 256, 320
    g = torch.Generator(device='cuda').manual_seed(3)
    img = torch.randn(B, 3, ph, pw, device='cuda', generator=g)
    gts, gls = synth.train_targets(11, B, ph, pw, max_gt=5)
    gtb = [torch.from_numpy(x).cuda() for x in gts]
    gtl = [torch.from_numpy(x).cuda() for x in gls]
    metas = [synth.img_meta(ph, pw, ph, pw) for _ in range(B)]
    out = {}
    for mode in (True, False):
        model.bbox_head.train_winograd = mode
        model.zero_grad()
        loss, lv = parse_losses(model(img, metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl))
        loss.backward()
        out[mode] = (float(loss), {n: p.grad.clone() for n, p in model.named_parameters()
                                   if p.grad is not None})
    (la, ga), (lb, gb) = out[True], out[False]
    assert abs(la - lb) <= 1e-5 * abs(lb)
    assert set(ga) == set(gb)
    for n in gb:
        assert _rel2(ga[n], gb[n]) < 1e-2, (n, _rel2(ga[n], gb[n]))


@pytest.mark.parametrize('cout,cin,cl', [(256, 256, False), (128, 64, True), (48, 256, False), (512, 512, True)])
def test_weight_transform_kernels_vs_float64_einsum(cout, cin, cl):
    """csrc/trainops.hip: U = G w G^T, its flipped / transposed variant for the input gradient,
    and the adjoint G^T dU G, against float64 einsums rounded once"""
    from iouaware import winograd as W
    from iouaware import winograd_train as WT
    g = torch.Generator(device='cuda').manual_seed(cout + cin)
    w = torch.randn(cout, cin, 3, 3, device='cuda', generator=g)
    if cl:
        w = w.contiguous(memory_format=torch.channels_last)
    def close(a, b):                  # both round a float64 result once; the sums differ in order
        return a.shape == b.shape and float((a - b).abs().max()) <= 2e-7 * float(b.abs().max())
    assert close(WT.transform_weight(w), W.transform_weight(w))
    assert close(WT.transform_weight(w, adjoint=True),
                 W.transform_weight(w.flip(2, 3).transpose(0, 1)))
    du = torch.randn(36, cin, cout, device='cuda', generator=g)
    G = torch.from_numpy(W._G).cuda()
    ref = torch.einsum('ik,ijco,jl->ockl', G, du.reshape(6, 6, cin, cout).double(), G)
    got = WT.untransform_weight_grad(du)
    assert got.shape == (cout, cin, 3, 3)
    assert float((got.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


@pytest.mark.parametrize('shape', [(2, 64, 33, 47), (3, 720, 13, 21), (2, 2048, 8, 11), (1, 36, 5, 7),
                                   (4, 256, 100, 168), (2, 1280, 9, 9)])
@pytest.mark.parametrize('relu', [True, False])
def test_relu_backward_and_bias_gradient_one_pass(shape, relu):
    from iouaware import winograd_train as WT
    g = torch.Generator(device='cuda').manual_seed(shape[1])
    dy = torch.randn(shape, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    y = torch.randn(shape, device='cuda', generator=g).relu().contiguous(memory_format=torch.channels_last)
    got, db = WT.relu_bwd_bias_grad(dy, y if relu else None, True)
    ref = dy * (y > 0) if relu else dy
    assert torch.equal(got, ref)
    rs = ref.double().sum((0, 2, 3))
    assert float((db.double() - rs).abs().max()) <= 1e-5 * float(ref.abs().double().sum((0, 2, 3)).max())
    got2, none = WT.relu_bwd_bias_grad(dy, y if relu else None, False)
    assert none is None and torch.equal(got2, ref)
