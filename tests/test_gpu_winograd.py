"""GPU (MI355X): the Winograd F(4x4,3x3) path (csrc/wino.hip + batched GEMM) against direct
convolutions (MIOpen) and against an fp64 convolution: same fp32 arithmetic class, sums
reassociated -> agreement to ~1e-5 of the activation scale, far inside the 1e-4 bar."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _err(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize('B,cin,cout,h,w', [(2, 64, 64, 17, 23), (1, 256, 256, 100, 168),
                                            (3, 128, 36, 8, 8), (2, 32, 720, 5, 7),
                                            (2, 256, 256, 1, 2), (1, 16, 16, 4, 4)])
def test_single_conv_matches_direct(B, cin, cout, h, w):
    from iouaware.winograd import WinogradConv3x3
    g = torch.Generator(device='cuda').manual_seed(h * w)
    x = _cl(torch.randn(B, cin, h, w, device='cuda', generator=g))
    wt = torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device='cuda', generator=g)
    for relu in (False, True):
        y = WinogradConv3x3(wt, bias, relu)(x)
        assert y.shape == (B, cout, h, w) and y.is_contiguous(memory_format=torch.channels_last)
        ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
        if relu:
            ref = ref.clamp(min=0)
        direct = F.conv2d(x, wt, bias, padding=1)
        direct = direct.clamp(min=0) if relu else direct
        assert _err(y, ref) < 2e-5, (_err(y, ref), _err(direct, ref))


def test_head_matches_module_forward_and_detections():
    """whole head (all levels, both towers, three outputs) vs the nn.Module convolutions, and the
    detections that come out of the HIP post-processing"""
    import iouaware
    from iouaware.config import ConfigDict
    from iouaware.fuse import fuse_inference, unfuse_inference
    from test_host_model import R50_MODEL, TEST_CFG
    import synth
    torch.manual_seed(3)
    m = iouaware.build_detector(ConfigDict(R50_MODEL), test_cfg=ConfigDict(TEST_CFG)).eval()
    # trained-like weights with wide score gaps (tests/synth.py, the E2E fixtures' scheme): the
    # default init (std 0.01) would make every logit nearly constant and every comparison a tie
    with torch.no_grad():
        synth.e2e_fill_state(m.state_dict(), 5)
    m = m.cuda()
    m = m.to(memory_format=torch.channels_last)
    img = _cl(torch.randn(2, 3, 224, 288, device='cuda'))
    metas = [synth.img_meta(220, 280, 224, 288) for _ in range(2)]
    with torch.no_grad():
        feats = m.extract_feat(img)
        ref = m.bbox_head(feats)
        fuse_inference(m, winograd=True)
        assert m.bbox_head._ia_wino.usable(feats)
        got = m.bbox_head(feats)
        for name, rs, gs in zip(('cls', 'reg', 'iou'), ref, got):
            for r, g in zip(rs, gs):
                assert g.shape == r.shape and g.is_contiguous(memory_format=torch.channels_last)
                e = float((g - r).abs().max() / r.abs().max())
                assert e < 5e-5, (name, tuple(r.shape), e)
        dets = m.simple_test_batch(img, metas, rescale=True)
        unfuse_inference(m)
        assert not hasattr(m.bbox_head, '_ia_wino')
        dets0 = m.simple_test_batch(img, metas, rescale=True)
    from test_gpu_e2e import _match_sets
    for d, d0 in zip(dets, dets0):          # the same detections, class by class, within 1e-4
        matched, total, _, _ = _match_sets(d0, d)
        assert total > 0 and sum(len(x) for x in d) == total and matched >= total - 1, (matched, total)


def test_whole_network_winograd_matches_module_path():
    """fuse_inference(winograd=True): Bottleneck 3x3 convs (BN folded into the weights), FPN output
    convs and the head -- against the plain nn.Module forward of the same weights"""
    import iouaware
    from iouaware.config import ConfigDict
    from iouaware.fuse import fuse_inference, unfuse_inference
    import bench
    torch.manual_seed(0)
    m = iouaware.build_detector(ConfigDict(bench.MODEL), test_cfg=ConfigDict(bench.TEST_CFG)).cuda().eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_(0, 0.1)
        for p in m.bbox_head.parameters():
            if p.dim() == 4:
                p.normal_(0, (2.0 / (9 * p.shape[1])) ** 0.5)
    m = m.to(memory_format=torch.channels_last)
    x = _cl(torch.randn(2, 3, 224, 288, device='cuda'))
    with torch.no_grad():
        ref = m.forward_head(x)
        n = fuse_inference(m, winograd=True)
        wino_blocks = [b for b in m.backbone.modules() if getattr(b, '_ia_fused', {}).get('wino2')]
        assert len(wino_blocks) == 13                      # 16 bottlenecks - 3 with a stride-2 conv2
        assert sum(1 for c in m.neck.fpn_convs if c._ia_fused.get('wino')) == 3
        out = m.forward_head(x)
        unfuse_inference(m)
    for a, b in zip(ref, out):
        for u, v in zip(a, b):
            assert v.is_contiguous(memory_format=torch.channels_last)
            e = float((u - v).abs().max() / u.abs().max())
            assert e < 1e-4, e


@pytest.mark.parametrize('k,n,batch,rows', [(64, 64, 36, 4200), (128, 128, 36, 4111), (256, 48, 36, 5000)])
def test_batched_gemm_stream_matches_fp64_and_library(k, n, batch, rows):
    """the HBM-bound Winograd-domain products on the streaming MFMA kernel (one grid row per matrix,
    weights in LDS) against fp64 and against the library's batched GEMM; rows % 16 != 0"""
    from iouaware import winograd as wg
    g = torch.Generator(device='cuda').manual_seed(k + n)
    v = torch.randn(batch, rows, k, device='cuda', generator=g)
    u = torch.randn(batch, k, n, device='cuda', generator=g) * 0.1
    out = torch.full((batch, rows, n), float('nan'), device='cuda')
    if (k, n) in wg._STREAM_BMM_SHAPES:
        wg.batched_gemm(v, u, out)                          # routed to ia_batched_gemm_stream
    else:                                                   # instantiated, not routed (slower than the library)
        from iouaware import _lib
        from iouaware.ops import _ptr, _stream
        _lib.check(_lib.lib().ia_batched_gemm_stream(_ptr(v), _ptr(u), _ptr(out), batch, rows, k, n, _stream()), 'bgs')
    want = torch.bmm(v.double(), u.double())
    assert float((out.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    wg.STREAM_BMM = False
    try:
        lib = wg.batched_gemm(v, u, torch.empty_like(out))
    finally:
        wg.STREAM_BMM = True
    assert float((out - lib).abs().max()) < 1e-5 * float(lib.abs().max())


def test_winograd_plans_are_a_bounded_lru():
    """evaluation meets many pad shapes: the per-layer plan caches (the head's plan owns two
    activation sets) keep the most recently used few and give the same result when a shape
    comes back after its plan was evicted"""
    from iouaware import winograd as W
    g = torch.Generator(device='cuda').manual_seed(3)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).cuda()
    layer = W.WinogradConv3x3(conv.weight, conv.bias, relu=True)
    xs = [torch.randn(1, 64, 8 + 4 * i, 12, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
          for i in range(W._PLAN_ENTRIES + 3)]
    with torch.no_grad():
        first = layer(xs[0]).clone()
        for x in xs[1:]:
            layer(x)
        assert len(layer._plans) == W._PLAN_ENTRIES
        again = layer(xs[0])
    assert torch.equal(first, again)
    assert len(layer._plans) == W._PLAN_ENTRIES
