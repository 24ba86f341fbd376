"""GPU: BASELINE config 5 in miniature -- whole training iterations of IoU-aware RetinaNet
R-50-FPN with the HIP loss kernels (forward_train -> losses -> backward -> clip -> SGD)."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def test_training_iterations_run_and_learn():
    import iouaware
    from iouaware.config import ConfigDict
    from iouaware.train import build_optimizer, parse_losses, train_step
    import bench
    from test_host_targets import TRAIN_CFG
    torch.manual_seed(0)
    model = iouaware.build_detector(ConfigDict(bench.MODEL), train_cfg=TRAIN_CFG,
                                    test_cfg=ConfigDict(bench.TEST_CFG)).cuda()
    model.train()
    opt = build_optimizer(model, dict(type='SGD', lr=0.005, momentum=0.9, weight_decay=0.0001))
    B, ph, pw = 2, 256, 320
    g = torch.Generator(device='cuda').manual_seed(3)
    img = torch.randn(B, 3, ph, pw, device='cuda', generator=g)
    gts, gls = synth.train_targets(11, B, ph, pw, max_gt=5)
    gtb = [torch.from_numpy(x).cuda() for x in gts]
    gtl = [torch.from_numpy(x).cuda() for x in gls]
    metas = [synth.img_meta(ph, pw, ph, pw) for _ in range(B)]
    frozen = model.backbone.layer1[0].conv1.weight.detach().clone()
    head_w = model.bbox_head.retina_cls.weight.detach().clone()
    hist = []
    for _ in range(6):
        lv = train_step(model, opt, img, metas, gtb, gtl, grad_clip=dict(max_norm=35, norm_type=2))
        assert set(lv) == {'loss_cls', 'loss_bbox', 'losses_iou', 'loss'}
        assert all(np.isfinite(v) for v in lv.values())
        assert abs(lv['loss'] - (lv['loss_cls'] + lv['loss_bbox'] + lv['losses_iou'])) < 1e-4 * lv['loss']
        hist.append(lv['loss'])
    assert hist[-1] < hist[0], hist                       # same batch 6 times: the loss goes down
    assert torch.equal(model.backbone.layer1[0].conv1.weight, frozen)      # frozen_stages=1
    assert not torch.equal(model.bbox_head.retina_cls.weight, head_w)
    assert not model.backbone.bn1.training                                  # norm_eval
    # every trainable parameter received a gradient (incl. retina_iou and the reg tower via the
    # attached IoU target)
    missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing


def test_build_optimizer_fused_sgd_equals_foreach_sgd():
    """build_optimizer picks torch's fused SGD for CUDA parameters: the same update as the default
    implementation (momentum, weight decay), to fp32 rounding"""
    import torch.nn as nn
    from iouaware.train import build_optimizer
    torch.manual_seed(0)
    a = nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.Conv2d(8, 4, 1)).cuda()
    b = nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.Conv2d(8, 4, 1)).cuda()
    b.load_state_dict(a.state_dict())
    cfg = dict(type='SGD', lr=0.05, momentum=0.9, weight_decay=0.0001)
    oa = build_optimizer(a, cfg)
    ob = build_optimizer(b, dict(cfg, foreach=True))
    assert oa.defaults.get('fused') is True and not ob.defaults.get('fused')
    x = torch.randn(4, 3, 16, 16, device='cuda')
    for _ in range(3):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            m(x).square().mean().backward()
            o.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert float((p - q).abs().max()) <= 1e-6 * max(float(q.abs().max()), 1e-12)


def test_ddp_wrapped_fused_training_step_matches_unwrapped():
    """`bench.py --config r50-train` at N > 1 wraps the model in DistributedDataParallel (RCCL).  A
    one-rank RCCL group exercises the same reducer / bucket hooks on this library's autograd nodes
    (folded-BN GEMM / Winograd convolutions, fused head losses): the wrapped step must produce the
    gradients and the update of the plain step."""
    import copy
    import os
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    import iouaware
    from iouaware.config import ConfigDict
    from iouaware.fuse import fuse_inference
    from iouaware.train import build_optimizer, train_step
    import bench
    from test_host_targets import TRAIN_CFG
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(bench.free_port()))
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        a = iouaware.build_detector(ConfigDict(bench.MODEL), train_cfg=TRAIN_CFG,
                                    test_cfg=ConfigDict(bench.TEST_CFG)).cuda().train()
        b = copy.deepcopy(a)
        B, ph, pw = 2, 256, 320
        g = torch.Generator(device='cuda').manual_seed(3)
        img = torch.randn(B, 3, ph, pw, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
        gts, gls = synth.train_targets(11, B, ph, pw, max_gt=5)
        gtb = [torch.from_numpy(x).cuda() for x in gts]
        gtl = [torch.from_numpy(x).cuda() for x in gls]
        metas = [synth.img_meta(ph, pw, ph, pw) for _ in range(B)]
        out = []
        for m, wrap in ((a, False), (b, True)):
            fuse_inference(m, winograd=True, train=True)
            m = m.to(memory_format=torch.channels_last)
            opt = build_optimizer(m, dict(type='SGD', lr=0.005, momentum=0.9, weight_decay=0.0001))
            net = DistributedDataParallel(m, device_ids=[0], broadcast_buffers=False) if wrap else m
            for _ in range(2):
                lv = train_step(net, opt, img, metas, gtb, gtl, grad_clip=dict(max_norm=35, norm_type=2))
            out.append((lv, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None},
                        {n: p.detach().clone() for n, p in m.named_parameters()}))
        (la, ga, pa), (lb, gb, pb) = out
        assert set(ga) == set(gb) and len(ga) > 100
        assert abs(la['loss'] - lb['loss']) < 1e-4 * abs(la['loss'])
        for n in ga:                              # library GEMMs / convolutions are not bit-reproducible
            d = float((ga[n] - gb[n]).norm()), float(ga[n].norm())
            assert d[0] <= 2e-3 * d[1] + 1e-7, (n, d)
        for n in pa:
            assert float((pa[n] - pb[n]).abs().max()) <= 1e-4 * float(pa[n].abs().max()) + 1e-7, n
    finally:
        dist.destroy_process_group()
