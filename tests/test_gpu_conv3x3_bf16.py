"""GPU: the bf16 3x3 implicit-GEMM convolution (csrc/conv3x3_bf16.hip) against torch.  Reference:
the same convolution in fp64 on the bf16-rounded inputs; the kernel accumulates in fp32 and rounds
the result to bf16 once, so |error| <= 2^-9 |y| (one bf16 rounding) + the fp32 accumulation noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,H,W,ci,co,relu,bias', [
    (1, 7, 11, 32, 256, False, False),        # one partial tile
    (2, 13, 21, 64, 256, True, True),
    (1, 25, 42, 256, 256, True, True),        # P5
    (2, 50, 84, 256, 256, True, True),        # P4: 9 x 28 tiles
    (1, 100, 168, 256, 256, False, True),     # P3: 10 x 24 tiles
    (1, 33, 17, 96, 512, True, True),         # two column blocks, Cin not a power of two
    (8, 100, 168, 64, 512, True, True),       # two column tiles, many workgroups
    (2, 40, 56, 64, 64, True, True),          # ResNet stage 1: the (2, 2, 2) variant, second wavefront column all padding
    (2, 37, 53, 128, 128, True, True),        # ResNet stage 2: (2, 2, 2), 128-pixel tiles with overhang
    (1, 25, 42, 512, 512, True, True),        # ResNet stage 4
    (1, 20, 30, 64, 96, False, True),         # 64 < Cout <= 128, partial second column
    (16, 100, 168, 256, 256, True, True),     # config 3's head tower on P3 at its batch: the (4, 1, 4) variant with the fragment ring
    (3, 61, 47, 160, 320, True, True),        # odd sizes, five chunks, a partial second column tile
])
def test_conv3x3_bf16_matches_fp64_convolution(B, H, W, ci, co, relu, bias):
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(H * W + ci)
    x = torch.randn(B, ci, H, W, device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (9 * ci)) ** 0.5).to(torch.bfloat16)
    b = torch.randn(co, device='cuda', generator=g) if bias else None
    wp = ops.conv3x3_bf16_pack(w)
    y = ops.conv3x3_bf16(x, wp, b, co, relu=relu)
    assert y.shape == (B, co, H, W) and y.dtype == torch.bfloat16
    assert y.is_contiguous(memory_format=torch.channels_last)
    want = torch.nn.functional.conv2d(x.double(), w.double(), b.double() if bias else None, 1, 1)
    if relu:
        want = want.clamp(min=0)
    err = (y.double() - want).abs()
    tol = 2.0 ** -8 * want.abs() + 2e-3 * float(want.abs().max()) * 2.0 ** -8
    assert bool((err <= tol).all()), (float(err.max()), float(want.abs().max()))
    # and the framework's own bf16 convolution is not closer than this kernel by more than rounding
    eager = torch.nn.functional.conv2d(x, w.contiguous(memory_format=torch.channels_last), b.to(torch.bfloat16) if bias else None, 1, 1)
    if relu:
        eager = eager.clamp(min=0)
    e_eager = float((eager.double() - want).pow(2).mean().sqrt())
    e_mine = float((y.double() - want).pow(2).mean().sqrt())
    assert e_mine <= 1.2 * e_eager + 1e-6, (e_mine, e_eager)


def test_conv3x3_bf16_levels_groups_slices_and_odd_widths():
    """one launch over five pyramid levels and two groups whose tensors are channel halves of
    512-channel activations (the cls / reg towers), and a 720-channel output (three column tiles,
    the last one partial) -- against fp64 convolutions of the bf16-rounded inputs"""
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(11)
    # (the 1 x 1 and 1 x 2 maps: P6 / P7 of a 64 x 64 input -- a channel slice of a 1 x 1 map has the image
    # stride as its pixel stride; found by tools/fuzz_fused_model.py)
    sizes = [(28, 40), (14, 20), (7, 10), (1, 2), (1, 1)]
    B, F = 3, 256
    cl = torch.channels_last
    acts = [torch.randn(B, 2 * F, h, w, device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
            for h, w in sizes]
    outs = [torch.zeros_like(a) for a in acts]
    w2 = (torch.randn(2 * F, F, 3, 3, device='cuda', generator=g) * 0.03).to(torch.bfloat16)
    b2 = torch.randn(2 * F, device='cuda', generator=g)
    wp = ops.conv3x3_bf16_pack(w2, groups=2)
    ops.conv3x3_bf16_levels([[a[:, :F] for a in acts], [a[:, F:] for a in acts]], wp, b2, F,
                            [[o[:, :F] for o in outs], [o[:, F:] for o in outs]], relu=True)
    for a, o in zip(acts, outs):
        for k in range(2):
            want = torch.nn.functional.conv2d(a[:, k * F:(k + 1) * F].double(), w2[k * F:(k + 1) * F].double(),
                                              b2[k * F:(k + 1) * F].double(), 1, 1).clamp(min=0)
            err = (o[:, k * F:(k + 1) * F].double() - want).abs()
            assert bool((err <= 2.0 ** -8 * want.abs() + 1e-5 * float(want.abs().max())).all()), float(err.max())
    # 720 output channels from the cls half of the activations
    wc = (torch.randn(720, F, 3, 3, device='cuda', generator=g) * 0.03).to(torch.bfloat16)
    bc = torch.randn(720, device='cuda', generator=g)
    wpc = ops.conv3x3_bf16_pack(wc)
    cls = [torch.full((B, 720, h, w), 7.0, device='cuda').to(torch.bfloat16).contiguous(memory_format=cl) for h, w in sizes]
    ops.conv3x3_bf16_levels([[a[:, :F] for a in acts]], wpc, bc, 720, [cls], relu=False)
    for a, o in zip(acts, cls):
        want = torch.nn.functional.conv2d(a[:, :F].double(), wc.double(), bc.double(), 1, 1)
        err = (o.double() - want).abs()
        assert bool((err <= 2.0 ** -8 * want.abs() + 1e-5 * float(want.abs().max())).all()), float(err.max())


@pytest.mark.parametrize('seed', range(12))
def test_conv3x3_bf16_random_shapes(seed):
    """seeded random maps / channel counts / batches through every variant the launcher picks (tile
    shapes with wraps inside a 32-pixel block, narrow maps, partial column tiles, Cout from 2 to 600,
    Cin in 32-channel chunks): against the fp64 convolution of the bf16-rounded operands"""
    import numpy as np
    from iouaware import ops
    rs = np.random.RandomState(1000 + seed)
    B = int(rs.randint(1, 5))
    H, W = int(rs.randint(1, 70)), int(rs.randint(1, 90))
    ci = 32 * int(rs.randint(1, 9))
    co = 2 * int(rs.choice([1, 8, 17, 32, 45, 64, 100, 128, 150, 256, 300]))
    relu, bias = bool(rs.randint(0, 2)), bool(rs.randint(0, 2))
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.randn(B, ci, H, W, device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (9 * ci)) ** 0.5).to(torch.bfloat16)
    b = torch.randn(co, device='cuda', generator=g) if bias else None
    y = ops.conv3x3_bf16(x, ops.conv3x3_bf16_pack(w), b, co, relu=relu)
    want = torch.nn.functional.conv2d(x.double(), w.double(), b.double() if bias else None, 1, 1)
    if relu:
        want = want.clamp(min=0)
    err = (y.double() - want).abs()
    tol = 2.0 ** -8 * want.abs() + 2e-3 * float(want.abs().max()) * 2.0 ** -8
    assert bool((err <= tol).all()), (B, H, W, ci, co, float(err.max()), float(want.abs().max()))
