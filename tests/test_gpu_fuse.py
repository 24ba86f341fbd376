"""GPU: the conv-epilogue kernel (csrc/elementwise.hip) against plain torch, and the fused
inference forward against the unfused model (same weights)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(2, 64, 37, 53), (3, 16, 8, 12), (1, 7, 1, 1), (2, 256, 100, 168)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_channel_affine_act_matches_torch(shape, dtype):
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(shape, device='cuda', generator=g).to(dtype)
    r = torch.randn(shape, device='cuda', generator=g).to(dtype)
    C = shape[1]
    s, b, rs, rb = [torch.randn(C, device='cuda', generator=g) for _ in range(4)]
    v = lambda t: t.view(1, C, 1, 1)          # noqa: E731
    xf, rf = x.float(), r.float()
    for relu in (False, True):
        want = xf * v(s) + v(b)
        got = ops.channel_affine_act_(x.clone(), s, b, relu=relu).float()
        w = want.clamp(min=0) if relu else want
        assert torch.equal(got, w.to(dtype).float())
        want = (xf * v(s) + v(b)) + (rf * v(rs) + v(rb))
        got = ops.channel_affine_act_(x.clone(), s, b, residual=r, res_scale=rs, res_shift=rb,
                                      relu=relu).float()
        w = want.clamp(min=0) if relu else want
        assert torch.equal(got, w.to(dtype).float())
        got = ops.channel_affine_act_(x.clone(), None, b, residual=r, relu=relu).float()
        want = (xf * 1.0 + v(b)) + rf
        w = want.clamp(min=0) if relu else want
        assert torch.equal(got, w.to(dtype).float())


@pytest.mark.parametrize('backbone', [dict(), dict(type='ResNeXt', depth=50, groups=32, base_width=4)])
def test_fused_inference_matches_unfused(backbone):
    import iouaware
    from iouaware.config import ConfigDict
    from iouaware.fuse import fuse_inference, unfuse_inference
    import bench
    cfg = ConfigDict(bench.MODEL)
    cfg.backbone.update(backbone)
    torch.manual_seed(0)
    m = iouaware.build_detector(cfg, test_cfg=ConfigDict(bench.TEST_CFG)).cuda().eval()
    with torch.no_grad():           # make BN statistics / residual branches non-trivial
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_(0, 0.1)
    x = torch.randn(2, 3, 224, 288, device='cuda')
    with torch.no_grad():
        ref = m.forward_head(x)
        assert fuse_inference(m) >= 33
        out = m.forward_head(x)
        unfuse_inference(m)
        again = m.forward_head(x)
    for a, b, c in zip(ref, out, again):
        for u, v, w in zip(a, b, c):
            scale = float(u.abs().max())
            assert float((u - v).abs().max()) <= 2e-4 * max(scale, 1.0)
            # (MIOpen may pick different conv algorithms between calls: no bit-equality here)
            assert float((u - w).abs().max()) <= 2e-4 * max(scale, 1.0)
    # training / grad mode keeps the ordinary path
    fuse_inference(m)
    m.train()
    y = m.forward_head(x)
    assert y[0][0].requires_grad


@pytest.mark.parametrize('shape', [(2, 720, 25, 42), (1, 9, 7, 11), (3, 36, 13, 21), (2, 64, 64, 64)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_nhwc_to_nchw_and_channels_last_epilogue(shape, dtype):
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(2)
    x = torch.randn(shape, device='cuda', generator=g).to(dtype)
    xcl = x.contiguous(memory_format=torch.channels_last)
    out = ops.to_nchw(xcl)
    assert out.is_contiguous() and torch.equal(out, x)
    if shape[1] % 8 == 0:
        C = shape[1]
        s, b = torch.randn(C, device='cuda', generator=g), torch.randn(C, device='cuda', generator=g)
        r = torch.randn(shape, device='cuda', generator=g).to(dtype)
        a = ops.channel_affine_act_(x.clone(), s, b, residual=r, relu=True)
        c = ops.channel_affine_act_(xcl.clone(memory_format=torch.channels_last), s, b,
                                    residual=r.contiguous(memory_format=torch.channels_last),
                                    relu=True)
        assert c.is_contiguous(memory_format=torch.channels_last) and torch.equal(a, c)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 64, 400, 672), (1, 64, 37, 53), (3, 8, 5, 4), (2, 16, 1, 1)])
def test_affine_relu_maxpool_matches_torch(shape, dtype):
    """bf16: fp32 arithmetic, ONE rounding at the end == eager's affine (rounded) -> ReLU -> max-pool
    on bf16 values, because rounding and ReLU are monotonic"""
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(*shape, device='cuda', generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    s = torch.randn(shape[1], device='cuda', generator=g)        # negative scales included
    t = torch.randn(shape[1], device='cuda', generator=g)
    got = ops.affine_relu_maxpool(x, s, t)
    aff = (x.float() * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)).to(dtype)
    want = torch.nn.functional.max_pool2d(torch.relu(aff), 3, 2, 1)
    assert got.dtype == dtype and got.shape == want.shape
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)


def test_linear_bias_act_matches_conv():
    """hipBLASLt 1x1-conv GEMM with bias / residual / ReLU in the epilogue vs conv2d + torch ops"""
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(2)
    x = torch.randn(2, 64, 50, 84, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(256, 64, 1, 1, device='cuda', generator=g) * 0.1
    b = torch.randn(256, device='cuda', generator=g)
    r = torch.randn(2, 256, 50, 84, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    w_kn = w.view(256, 64).t().contiguous()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double())
    for res, relu in ((None, False), (None, True), (r, True), (r, False)):
        want = ref + (res.double() if res is not None else 0)
        want = want.clamp(min=0) if relu else want
        got = ops.linear_bias_act(x, w_kn, b, residual=res, relu=relu)
        assert got.is_contiguous(memory_format=torch.channels_last)
        assert float((got.double() - want).abs().max()) < 1e-4 * float(want.abs().max())
    got = ops.linear_bias_act(x, w_kn, None)
    assert float((got.double() - (ref - b.double().view(1, -1, 1, 1))).abs().max()) < 1e-4 * float(ref.abs().max())


def test_upsample2x_add_matches_torch():
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(5)
    for (B, C_, Hc, Wc) in [(2, 256, 25, 42), (1, 8, 1, 1), (3, 64, 7, 5)]:
        fine = torch.randn(B, C_, 2 * Hc, 2 * Wc, device='cuda', generator=g).contiguous(
            memory_format=torch.channels_last)
        coarse = torch.randn(B, C_, Hc, Wc, device='cuda', generator=g).contiguous(
            memory_format=torch.channels_last)
        want = fine + torch.nn.functional.interpolate(coarse, scale_factor=2, mode='nearest')
        fb, cb = fine.to(torch.bfloat16), coarse.to(torch.bfloat16)          # still channels-last
        want_b = fb + torch.nn.functional.interpolate(cb, scale_factor=2, mode='nearest')
        got = ops.upsample2x_add_(fine, coarse)
        assert got.data_ptr() == fine.data_ptr() and torch.equal(got, want)
        assert torch.equal(ops.upsample2x_add_(fb, cb), want_b)


def test_linear_bias_act_bf16():
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(4)
    x = torch.randn(2, 256, 25, 42, device='cuda', generator=g).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    w = (torch.randn(64, 256, device='cuda', generator=g) * 0.06).to(torch.bfloat16)
    b = torch.randn(64, device='cuda', generator=g)
    r = torch.randn(2, 64, 25, 42, device='cuda', generator=g).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    got = ops.linear_bias_act(x, w.t().contiguous(), b, residual=r, relu=True)
    assert got.dtype == torch.bfloat16 and got.is_contiguous(memory_format=torch.channels_last)
    ref = torch.einsum('bkhw,nk->bnhw', x.double(), w.double()) + b.double().view(1, -1, 1, 1) + r.double()
    ref = ref.clamp(min=0)
    assert float((got.double() - ref).abs().max()) <= 1e-2 * float(ref.abs().max())    # bf16 output


def test_fused_bf16_network_matches_unfused():
    """BASELINE config 3 dtype: the fused path (bf16 hipBLASLt 1x1 GEMMs, bf16 epilogues) vs the
    plain bf16 modules, relative to the activation scale"""
    import iouaware
    from iouaware.config import ConfigDict
    from iouaware.fuse import fuse_inference, unfuse_inference
    import bench
    torch.manual_seed(0)
    m = iouaware.build_detector(ConfigDict(bench.MODEL), test_cfg=ConfigDict(bench.TEST_CFG)).cuda().eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_var.uniform_(0.5, 1.5); mod.weight.uniform_(0.5, 1.5)
        for p in m.bbox_head.parameters():
            if p.dim() == 4:
                p.normal_(0, (2.0 / (9 * p.shape[1])) ** 0.5)
    fuse_inference(m, winograd=True)
    m = m.to(memory_format=torch.channels_last).to(torch.bfloat16)
    x = torch.randn(2, 3, 224, 288, device='cuda').to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    with torch.no_grad():
        out = m.forward_head(x)
        unfuse_inference(m)
        ref = m.forward_head(x)
    for a, b in zip(ref, out):
        for u, v in zip(a, b):
            assert v.dtype == torch.bfloat16
            assert float((u.float() - v.float()).abs().max()) <= 0.08 * float(u.float().abs().max())


@pytest.mark.parametrize('k,n,res', [(64, 256, True), (64, 256, False), (256, 64, False), (64, 64, False)])
def test_conv1x1_stream_matches_fp64(k, n, res):
    """the streaming stage-1 1x1 kernel (weights in LDS, D^T = W^T X^T on v_mfma_f32_16x16x4_f32)
    against an fp64 product; partial last tile (rows % 16 != 0); and linear_bias_act routes the
    large stage-1 shapes to it"""
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(k + n)
    x = torch.randn(3, k, 37, 53, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(k, n, device='cuda', generator=g) * 0.1
    b = torch.randn(n, device='cuda', generator=g)
    r = torch.randn(3, n, 37, 53, device='cuda', generator=g).contiguous(memory_format=torch.channels_last) if res else None
    want = torch.einsum('bkhw,kn->bnhw', x.double(), w.double()) + b.double().view(1, -1, 1, 1)
    if res:
        want = want + r.double()
    for relu in (False, True):
        ref = want.clamp(min=0) if relu else want
        got = ops.conv1x1_stream(x, w, b, residual=r, relu=relu)
        assert got.is_contiguous(memory_format=torch.channels_last)
        assert float((got.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    got = ops.conv1x1_stream(x, w, None, relu=False)
    assert float((got.double() - (want - b.double().view(1, -1, 1, 1) - (r.double() if res else 0))).abs().max()) \
        < 1e-5 * float(want.abs().max())
    big = torch.randn(2, k, 200, 168, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    a = ops.linear_bias_act(big, w, b, relu=True)                   # routed to the streaming kernel
    ops.STREAM_1X1 = False
    try:
        lib = ops.linear_bias_act(big, w, b, relu=True)
    finally:
        ops.STREAM_1X1 = True
    assert float((a - lib).abs().max()) < 1e-5 * float(lib.abs().max())


@pytest.mark.parametrize('res', [True, False])
def test_conv1x1_chain_equals_the_two_kernels(res):
    """the block-boundary kernel (conv3 + add + ReLU -> y, next conv1 + ReLU -> h from the
    accumulators): the same bits as the two streaming kernels it replaces, and the fp64 products;
    partial last tile"""
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(3, 64, 37, 53, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 256, device='cuda', generator=g) * 0.1
    b = torch.randn(256, device='cuda', generator=g)
    w2 = torch.randn(256, 64, device='cuda', generator=g) * 0.05
    b2 = torch.randn(64, device='cuda', generator=g)
    r = torch.randn(3, 256, 37, 53, device='cuda', generator=g).contiguous(memory_format=torch.channels_last) if res else None
    y, h = ops.conv1x1_chain(x, w, b, r, w2, b2)
    y0 = ops.conv1x1_stream(x, w, b, residual=r, relu=True)
    h0 = ops.conv1x1_stream(y0, w2, b2, relu=True)
    assert y.is_contiguous(memory_format=torch.channels_last) and h.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, y0) and torch.equal(h, h0)
    want = torch.einsum('bkhw,kn->bnhw', x.double(), w.double()) + b.double().view(1, -1, 1, 1)
    if res:
        want = want + r.double()
    want = want.clamp(min=0)
    wh = (torch.einsum('bkhw,kn->bnhw', want, w2.double()) + b2.double().view(1, -1, 1, 1)).clamp(min=0)
    assert float((y.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    assert float((h.double() - wh).abs().max()) < 1e-5 * float(wh.abs().max())
    y, h = ops.conv1x1_chain(x, w, None, None, w2, None)            # no biases, no residual
    y0 = ops.conv1x1_stream(x, w, None, relu=True)
    assert torch.equal(y, y0) and torch.equal(h, ops.conv1x1_stream(y0, w2, None, relu=True))


def test_stage1_chained_boundaries_equal_unchained():
    """fuse._layer_forward: ResNet-50 stage 1 with the block boundaries chained (the tail of block i
    also produces conv1 of block i + 1) against the same blocks one kernel per convolution -- same
    bits; smaller inputs and the other stages take the unchained route"""
    from iouaware import ops
    from iouaware.backbones import ResNet
    from iouaware.fuse import fuse_inference
    torch.manual_seed(3)
    net = ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1, style='pytorch').cuda().eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    fuse_inference(net, winograd=True)
    x = torch.randn(2, 3, 736, 800, device='cuda').contiguous(memory_format=torch.channels_last)   # stage 1: 2 x 184 x 200
    calls = []
    orig = ops.conv1x1_chain
    ops.conv1x1_chain = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            a = net(x)
            n_chained = len(calls)
            ops.CHAIN_1X1 = False
            b = net(x)
            assert len(calls) == n_chained
            small = net(x[:, :, :256, :320].contiguous(memory_format=torch.channels_last))
            assert len(calls) == n_chained                  # below the streaming kernels' size: unchained
    finally:
        ops.CHAIN_1X1 = True
        ops.conv1x1_chain = orig
    assert n_chained == 2                                   # block 0 -> 1, block 1 -> 2
    assert len(small) == 4
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize('B,H,W', [(1, 37, 53), (2, 256, 320), (1, 7, 9), (2, 130, 515), (1, 800, 1344)])
def test_stem_conv_matches_fp64(B, H, W):
    """the 7x7 / stride-2 stem convolution on the fp32 MFMA kernel (csrc/stem.hip) against an fp64
    convolution: odd sizes, maps smaller than a workgroup tile, partial tiles in both directions,
    the benchmark's size"""
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(H * W)
    x = torch.randn(B, 3, H, W, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 3, 7, 7, device='cuda', generator=g) * 0.1
    y = ops.stem_conv(x, ops.stem_weight(w))
    # fp64 reference as unfold + matmul (the library's fp64 convolution takes minutes at these sizes)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    cols = torch.nn.functional.unfold(x.double().contiguous(), 7, padding=3, stride=2)        # (B, 147, Ho * Wo)
    want = torch.matmul(w.double().reshape(64, 147), cols).reshape(B, 64, Ho, Wo)
    assert y.shape == want.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert float((y.double() - want).abs().max()) < 1e-5 * float(want.abs().max())


def test_fused_stem_takes_the_own_kernel_and_matches_the_modules():
    """fuse_inference(winograd=True): ResNet._stem = own convolution kernel + the fused BN / ReLU /
    max-pool pass; against conv1 -> norm1 -> relu -> maxpool of the plain modules"""
    from iouaware import ops
    from iouaware.backbones import ResNet
    from iouaware.fuse import fuse_inference
    torch.manual_seed(5)
    net = ResNet(depth=50, num_stages=1, strides=(1,), dilations=(1,), out_indices=(0,)).cuda().eval()
    net.norm1.running_mean.normal_(0, 0.1); net.norm1.running_var.uniform_(0.5, 1.5)
    net.norm1.weight.data.uniform_(0.5, 1.5); net.norm1.bias.data.normal_(0, 0.1)
    x = torch.randn(2, 3, 203, 311, device='cuda').contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        want = net.maxpool(net.relu(net.norm1(net.conv1(x))))
        fuse_inference(net, winograd=True)
        assert 'stem_w' in net._ia_fused
        calls = []
        orig = ops.stem_conv
        ops.stem_conv = lambda *a: (calls.append(1), orig(*a))[1]
        try:
            got = net._stem(x)
        finally:
            ops.stem_conv = orig
    assert calls == [1]
    assert float((got - want).abs().max()) < 1e-5 * float(want.abs().max())


def test_conv1x1_wide_matches_fp64_and_library():
    """the stage-2 tail (128 -> 512 + residual + ReLU) on the streaming kernel's wide-output form
    (column blocks of 256 channels) against fp64 and the library GEMM it replaces; rows % 16 != 0"""
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(9)
    x = torch.randn(2, 128, 190, 173, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)   # 65 740 pixels
    w = torch.randn(128, 512, device='cuda', generator=g) * 0.08
    b = torch.randn(512, device='cuda', generator=g)
    r = torch.randn(2, 512, 190, 173, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    got = ops.linear_bias_act(x, w, b, residual=r, relu=True)            # routed to ia_conv1x1_wide
    ops.WIDE_1X1 = False
    try:
        lib = ops.linear_bias_act(x, w, b, residual=r, relu=True)
    finally:
        ops.WIDE_1X1 = True
    want = (torch.einsum('bkhw,kn->bnhw', x.double(), w.double()) + b.double().view(1, -1, 1, 1) + r.double()).clamp(min=0)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert float((got.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    assert float((got - lib).abs().max()) < 1e-5 * float(lib.abs().max())


@pytest.mark.parametrize('B,H,W', [(1, 37, 53), (2, 256, 320), (1, 7, 9), (1, 130, 515)])
def test_stem_conv_bf16_matches_fp64_of_the_rounded_operands(B, H, W):
    """the bf16 stem kernel (v_mfma_f32_16x16x16_bf16, weights in registers) against an fp64
    convolution of the bf16-rounded operands: fp32 accumulation, one rounding of the result"""
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(H + W)
    x = torch.randn(B, 3, H, W, device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 7, 7, device='cuda', generator=g) * 0.1).to(torch.bfloat16)
    y = ops.stem_conv_bf16(x, ops.stem_weight_bf16(w))
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    cols = torch.nn.functional.unfold(x.double().contiguous(), 7, padding=3, stride=2)
    want = torch.matmul(w.double().reshape(64, 147), cols).reshape(B, 64, Ho, Wo)
    assert y.shape == want.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    err = (y.double() - want).abs()
    tol = 2.0 ** -8 * want.abs() + 1e-5 * float(want.abs().max())
    assert bool((err <= tol).all()), (float(err.max()), float(want.abs().max()))
