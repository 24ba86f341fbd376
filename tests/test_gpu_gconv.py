"""GPU: the ResNeXt grouped 3x3 convolution on MFMA (csrc/gconv.hip, BASELINE config 4) against
F.conv2d: every group width of X-101-32x4d / 64x4d (4, 8, 16, 32 channels per group), stride 1 and
2, odd sizes, folded scale + bias + ReLU.  fp32 both sides, different summation order: 1e-5 of
the output scale."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('C,groups', [(256, 64), (512, 64), (1024, 64), (2048, 64), (128, 32),
                                      (1024, 32), (64, 4)])
@pytest.mark.parametrize('stride', [1, 2])
@pytest.mark.parametrize('hw', [(37, 53), (16, 16), (7, 11), (1, 1)])
def test_grouped_conv3x3_matches_conv2d(C, groups, stride, hw):
    from iouaware import ops
    g = torch.Generator(device='cuda').manual_seed(C + groups + stride)
    cg = C // groups
    H, W = hw
    x = torch.randn(2, C, H, W, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, cg, 3, 3, device='cuda', generator=g) * (1.0 / (9 * cg)) ** 0.5
    scale = torch.rand(C, device='cuda', generator=g) + 0.5
    bias = torch.randn(C, device='cuda', generator=g) * 0.1
    want = F.conv2d(x, w * scale.view(-1, 1, 1, 1), bias, stride=stride, padding=1, groups=groups)
    wp = ops.pack_grouped_weight(w, scale)
    for relu in (False, True):
        got = ops.grouped_conv3x3(x, wp, bias, groups, stride, relu=relu)
        ref = want.clamp(min=0) if relu else want
        assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
        err = float((got - ref).abs().max())
        assert err <= 1e-5 * max(1.0, float(ref.abs().max())), (C, groups, stride, hw, err)
    got = ops.grouped_conv3x3(x, ops.pack_grouped_weight(w), None, groups, stride)
    ref = F.conv2d(x, w, None, stride=stride, padding=1, groups=groups)
    assert float((got - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('C,H,W,stride', [
    (256, 200, 336, 1),                       # layer1 conv2
    (512, 200, 336, 2), (512, 100, 168, 1),   # layer2: first block (stride 2), the others
    (1024, 100, 168, 2), (1024, 50, 84, 1),   # layer3
    (2048, 50, 84, 2), (2048, 25, 42, 1),     # layer4
])
def test_grouped_conv3x3_at_the_benchmark_shapes(C, H, W, stride):
    """every conv2 shape of X-101-64x4d (BASELINE config 4) as `bench.py --config
    x101-64x4d` runs it -- batch 8, 800 x 1344 input, 64 groups -- against the library's grouped
    convolution (fp32 both sides, other summation order: 1e-5 of the output scale), plus the same
    bits from a second launch."""
    from iouaware import ops
    groups, B = 64, 8
    g = torch.Generator(device='cuda').manual_seed(C + H + stride)
    cg = C // groups
    x = torch.randn(B, C, H, W, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, cg, 3, 3, device='cuda', generator=g) * (1.0 / (9 * cg)) ** 0.5
    scale = torch.rand(C, device='cuda', generator=g) + 0.5
    bias = torch.randn(C, device='cuda', generator=g) * 0.1
    wp = ops.pack_grouped_weight(w, scale)
    got = ops.grouped_conv3x3(x, wp, bias, groups, stride, relu=True)
    again = ops.grouped_conv3x3(x, wp, bias, groups, stride, relu=True)
    assert torch.equal(got, again)
    want = F.conv2d(x, w * scale.view(-1, 1, 1, 1), bias, stride=stride, padding=1, groups=groups).clamp(min=0)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    err = float((got - want).abs().max())
    assert err <= 1e-5 * max(1.0, float(want.abs().max())), (C, H, W, stride, err)
    # the image borders and the last image of the batch in particular
    for sl in ((0, slice(None), 0), (B - 1, slice(None), -1), (B - 1, -1, slice(None))):
        b_, y_, x_ = sl
        assert float((got[b_, :, y_, x_] - want[b_, :, y_, x_]).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))


def test_grouped_conv3x3_rejects_what_it_does_not_cover():
    from iouaware import ops, _lib
    x = torch.randn(1, 96, 8, 8, device='cuda').contiguous(memory_format=torch.channels_last)
    with pytest.raises(_lib.IouAwareLibraryError):          # 12 channels per group
        ops.pack_grouped_weight(torch.randn(96, 12, 3, 3, device='cuda'))
    with pytest.raises(ValueError):                          # NCHW input
        ops.grouped_conv3x3(torch.randn(1, 64, 8, 8, device='cuda'), x, None, 4)
