"""GPU: the operator-level boundary accepts what the reference's operators accept (VERDICT r3
"missing" 2-4, SURVEY 8b "native operator level"):

  * `nms` on CPU tensors and ndarrays without device_id (nms_wrapper.py:27-45 -> nms_cpu.nms):
    staged to the device and back, same type / device as the input;
  * `nms` on float64 boxes (nms_cpu.cpp:63): fp64 kernels, pinned on the reference's own compiled
    op (tests/golden/nms_f64.npz) incl. the case where fp32 and fp64 disagree;
  * `multiclass_nms` on CPU tensors, and beyond the batched kernels' capacities (bbox_nms.py:33-56
    takes any n / max_num): 12 000 boxes, 27 519 survivors (tests/golden/mnms_big.npz);
  * `sigmoid_focal_loss` in half / bf16 / double storage (sigmoid_focal_loss_cuda.cu:128,166)."""
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def test_nms_accepts_cpu_tensors_and_ndarrays(golden_dir):
    from iouaware import nms_op
    f = np.load(os.path.join(golden_dir, 'nms.npz'))
    for i in range(int(f['num_cases'])):
        dets, thr, keep = f['dets_%d' % i], float(f['thr_%d' % i]), f['keep_%d' % i]
        t = torch.from_numpy(dets)                                  # CPU tensor
        out, inds = nms_op.nms(t, thr)
        assert not inds.is_cuda and inds.dtype == torch.int64 and not out.is_cuda
        assert np.array_equal(inds.numpy(), keep) and np.array_equal(out.numpy(), dets[keep])
        out, inds = nms_op.nms(dets, thr)                            # ndarray, no device_id
        assert isinstance(inds, np.ndarray) and isinstance(out, np.ndarray) and inds.dtype == np.int64
        assert np.array_equal(inds, keep) and np.array_equal(out, dets[keep])
        out, inds = nms_op.nms(dets, thr, device_id=0)               # ndarray + device_id
        assert isinstance(inds, np.ndarray) and np.array_equal(inds, keep)
        out, inds = nms_op.nms(t.cuda(), thr)                        # device tensor: stays there
        assert inds.is_cuda and np.array_equal(inds.cpu().numpy(), keep)


def test_nms_float64_matches_the_reference_op(golden_dir):
    from iouaware import nms_op
    f = np.load(os.path.join(golden_dir, 'nms_f64.npz'))
    for i in range(int(f['num_cases'])):
        dets, thr, keep = f['dets_%d' % i], float(f['thr_%d' % i]), f['keep_%d' % i]
        assert dets.dtype == np.float64
        out, inds = nms_op.nms(torch.from_numpy(dets).cuda(), thr)
        assert out.dtype == torch.float64 and np.array_equal(inds.cpu().numpy(), keep), i
        out, inds = nms_op.nms(dets, thr)                            # ndarray of doubles, staged
        assert out.dtype == np.float64 and np.array_equal(inds, keep), i
    d = f['edge_dets']
    for j in range(3):
        thr = float(f['edge_thr_%d' % j])
        for dt in (np.float64, np.float32):
            _, inds = nms_op.nms(torch.from_numpy(d.astype(dt)).cuda(), thr)
            assert np.array_equal(inds.cpu().numpy(), f['edge_keep_%d_%s' % (j, np.dtype(dt).name)]), (j, dt)
    # thresholds that are / are not exactly representable as a C float (nms_cpu.cpp:5): IoU exactly 0.5
    # and 0.25 against 0.5 / 0.25 (both types suppress); IoU 0.300000005 against float(0.3) =
    # 0.30000001192...: the double op keeps the box (a port comparing with double(0.3) would not)
    for j in range(3):
        dd, thr = f['edge2_dets_%d' % j], float(f['edge2_thr_%d' % j])
        for dt in (np.float64, np.float32):
            _, inds = nms_op.nms(torch.from_numpy(dd.astype(dt)).cuda(), thr)
            assert np.array_equal(inds.cpu().numpy(), f['edge2_keep_%d_%s' % (j, np.dtype(dt).name)]), (j, dt)
    assert f['edge2_keep_2_float64'].tolist() == [0, 1] and f['edge2_keep_2_float32'].tolist() == [0]
    # IoU exactly 1/3 against float(1/3): the two instantiations of the reference disagree, so do ours
    assert f['edge_keep_0_float64'].tolist() == [0, 1] and f['edge_keep_0_float32'].tolist() == [0]
    # ties in the scores: canonical order (index ascending), same bits twice
    rs = np.random.RandomState(3)
    xy = rs.uniform(0, 200, (3000, 2))
    t = np.concatenate([xy, xy + rs.uniform(5, 60, (3000, 2)), rs.randint(0, 7, (3000, 1)) / 7.0], 1)
    a = nms_op.nms(torch.from_numpy(t).cuda(), 0.5)[1]
    assert torch.equal(a, nms_op.nms(torch.from_numpy(t).cuda(), 0.5)[1])
    assert bool((a[1:] > a[:-1]).all())
    with pytest.raises(Exception, match='16384'):
        nms_op.nms(torch.zeros(16385, 5, dtype=torch.float64, device='cuda'), 0.5)


def test_multiclass_nms_on_cpu_tensors(golden_dir):
    from iouaware import nms_op
    f = np.load(os.path.join(golden_dir, 'mnms_quirk.npz'))
    boxes, scores = torch.from_numpy(f['boxes']), torch.from_numpy(f['scores'])
    cfg = dict(type='nms', iou_thr=float(f['iou_thr']))
    for name, mx in (('m1', -1), ('k20', 20), ('all', 1024)):
        b, l = nms_op.multiclass_nms(boxes, scores, float(f['score_thr']), cfg, mx)
        assert not b.is_cuda and not l.is_cuda and l.dtype == torch.long
        assert np.array_equal(l.numpy(), f['labels_' + name]) and np.array_equal(b.numpy(), f['bboxes_' + name])


def test_multiclass_nms_beyond_the_batched_capacities(golden_dir):
    from iouaware import nms_op, ops
    f = np.load(os.path.join(golden_dir, 'mnms_big.npz'))
    boxes, sc = synth.mnms_big_inputs()
    assert synth.checksum([boxes, sc]) == int(f['checksum'])
    assert boxes.shape[0] > ops._lib.IA_MAX_CANDIDATES
    cfg = dict(type='nms', iou_thr=float(f['iou_thr']))
    B, S = torch.from_numpy(boxes).cuda(), torch.from_numpy(sc).cuda()
    for name, mx in (('m1', -1), ('k3000', 3000), ('k100', 100)):
        b, l = nms_op.multiclass_nms(B, S, float(f['score_thr']), cfg, mx)
        assert b.is_cuda and l.dtype == torch.long
        assert np.array_equal(l.cpu().numpy(), f['labels_' + name].astype(np.int64)), name
        assert np.array_equal(b.cpu().numpy(), f['bboxes_' + name]), name
    # fewer boxes than the capacity but more survivors than the batched output holds (max_num = -1)
    sub = slice(0, 6000)
    b0, l0 = nms_op._multiclass_nms_per_class(B[sub], S[sub], float(f['score_thr']), 'nms', dict(iou_thr=0.5), -1)
    b1, l1 = nms_op.multiclass_nms(B[sub], S[sub], float(f['score_thr']), cfg)
    assert b0.shape[0] >= ops._lib.IA_MAX_PER_IMG and torch.equal(b0, b1) and torch.equal(l0, l1)
    # and inside the capacities the two routes agree bit for bit
    sub = slice(0, 1500)
    for mx in (50, 700):
        b0, l0 = nms_op._multiclass_nms_per_class(B[sub], S[sub], 0.2, 'nms', dict(iou_thr=0.5), mx)
        b1, l1 = nms_op.multiclass_nms(B[sub], S[sub], 0.2, cfg, mx)
        assert torch.equal(b0, b1) and torch.equal(l0, l1)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16, torch.float64, torch.float32])
def test_sigmoid_focal_loss_op_in_every_storage_type(golden_dir, dtype):
    """the op in the logits' own type: fp32 arithmetic (as the reference kernel's expf / powf /
    logf), ONE rounding to the storage type -- against the reference-pinned fp32 values of
    tests/golden/focal_op.npz on the type-rounded logits"""
    from iouaware.focal_op import sigmoid_focal_loss
    f = np.load(os.path.join(golden_dir, 'focal_op.npz'))
    x32 = torch.from_numpy(f['logits']).cuda()
    t = torch.from_numpy(f['targets']).cuda()
    up = torch.from_numpy(f['upstream']).cuda()
    x = x32.to(dtype)
    for gamma, alpha in f['params'].tolist():
        xr = x.float().clone().requires_grad_(True)                 # fp32 op on the rounded logits
        want = sigmoid_focal_loss(xr, t, gamma, alpha, 'none')
        (want * up.to(dtype).float()).sum().backward()
        xd = x.clone().requires_grad_(True)
        got = sigmoid_focal_loss(xd, t, gamma, alpha, 'none')
        assert got.dtype == dtype and xd.dtype == dtype
        (got * up.to(dtype)).sum().backward()
        assert xd.grad.dtype == dtype
        if dtype in (torch.float32, torch.float64):
            assert torch.equal(got.float(), want.detach()) and torch.equal(xd.grad.float(), xr.grad)
        else:
            assert torch.equal(got, want.detach().to(dtype))          # one rounding of the fp32 value
            assert torch.equal(xd.grad, xr.grad.to(dtype))
    # the reference kernel keeps its intermediates in scalar_t (sigmoid_focal_loss_cuda.cu:
    # 38-62: several half roundings for half, double arithmetic for double); this op computes in fp32
    # and rounds ONCE.  The deviation is pinned against the reference's own function evaluated in the
    # storage type (tests/golden/focal_op.npz, loss64 / loss16 keys): double: fp32 accuracy (2e-6
    # relative); half: within the scatter of the reference's own half roundings (4 half ulps)
    if dtype in (torch.float64, torch.float16):
        key = '64' if dtype == torch.float64 else '16'
        rel, absol = (2e-6, 2e-6) if dtype == torch.float64 else (4 * 2.0 ** -10, 2e-3)
        for gamma, alpha in f['params'].tolist():
            tag = 'g%g_a%g' % (gamma, alpha)
            xd = x.clone().requires_grad_(True)
            got = sigmoid_focal_loss(xd, t, gamma, alpha, 'none')
            (got * up.to(dtype)).sum().backward()
            for mine, ref in ((got.detach(), f['loss%s_%s' % (key, tag)]), (xd.grad, f['grad%s_%s' % (key, tag)])):
                mine, ref = mine.double().cpu().numpy(), ref.astype(np.float64)
                assert (np.abs(mine - ref) <= rel * np.abs(ref) + absol).all(), \
                    (tag, key, float(np.abs(mine - ref).max()))
    if dtype == torch.float32:                                        # and fp32 stays pinned on the reference
        tag = 'g2_a0.25'
        got = sigmoid_focal_loss(x32, t, 2.0, 0.25, 'none').cpu().numpy()
        assert np.abs(got - f['loss_' + tag]).max() <= 1e-5 * max(1.0, np.abs(f['loss_' + tag]).max())
