"""CPU: the pre-processing restatement (oracle/iouaware_oracle_preproc.c).
PARITY UNPINNED for the resize: cv2 / mmcv are third-party packages absent from the reference tree
and from this image, and the reference ships no fixture.  What can be checked here: the
fixed-point bilinear stays within one grey level of exact-arithmetic bilinear, the special cases
(copy, 2x reduction = 2x2 mean, constant images) are exact, the steps after the resize equal numpy's
fp32 arithmetic bit for bit, and the sizes follow mmcv's rules on the familiar COCO cases."""
import numpy as np


def _exact_bilinear(im, nh, nw):
    h, w = im.shape[:2]
    sx = (np.arange(nw) + 0.5) * (w / nw) - 0.5
    sy = (np.arange(nh) + 0.5) * (h / nh) - 0.5
    x0 = np.floor(sx).astype(int); fx = sx - x0
    y0 = np.floor(sy).astype(int); fy = sy - y0
    fx = np.where((x0 < 0) | (x0 >= w - 1), 0.0, fx)
    x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    f = im.astype(np.float64)
    top = f[y0c][:, x0c] * (1 - fx)[None, :, None] + f[y0c][:, x1c] * fx[None, :, None]
    bot = f[y1c][:, x0c] * (1 - fx)[None, :, None] + f[y1c][:, x1c] * fx[None, :, None]
    return top * (1 - fy)[:, None, None] + bot * fy[:, None, None]


def test_sizes_follow_mmcv(oracle_lib):
    assert oracle_lib.rescale_size(480, 640, (1333, 800))[:2] == (800, 1067)
    assert oracle_lib.rescale_size(427, 640, (1333, 800))[:2] == (800, 1199)
    assert oracle_lib.rescale_size(640, 480, (1333, 800)) == (1067, 800, 800 / 480)
    nh, nw, sf = oracle_lib.rescale_size(240, 320, (512, 384), keep_ratio=False)
    assert (nh, nw) == (384, 512) and sf.dtype == np.float32
    assert np.array_equal(sf, np.array([1.6, 1.6, 1.6, 1.6], np.float32))


def test_an_empty_rescaled_image_is_rejected(oracle_lib):
    """842 x 1 under keep_ratio: int(1 * 227 / 842 + 0.5) = 0 columns -- cv2.resize behind mmcv.imrescale
    (transforms.py:35) asserts a non-empty dsize, so the reference raises; found by tools/fuzz_preproc_soft.py
    seed 605084 (the oracle used to return a 227 x 0 image, the C-ABI answered IA_E_ARG)"""
    import pytest
    assert oracle_lib.rescale_size(842, 1, (172, 227))[:2] == (227, 0)
    with pytest.raises(ValueError, match='empty'):
        oracle_lib.image_transform(np.zeros((842, 1, 3), np.uint8), (172, 227), False, True, size_divisor=32)


def test_resize_close_to_exact_bilinear_and_special_cases(oracle_lib):
    rs = np.random.RandomState(0)
    for (h, w, nh, nw) in [(480, 640, 800, 1067), (600, 900, 200, 300), (33, 47, 567, 800),
                           (97, 131, 80, 211)]:
        im = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        r = oracle_lib.resize_bilinear_u8(im, nh, nw)
        assert np.abs(r.astype(np.float64) - _exact_bilinear(im, nh, nw)).max() < 1.0
    im = rs.randint(0, 256, (64, 96, 3)).astype(np.uint8)
    assert np.array_equal(oracle_lib.resize_bilinear_u8(im, 64, 96), im)          # dsize == ssize
    half = oracle_lib.resize_bilinear_u8(im, 32, 48).astype(np.int32)            # == INTER_AREA 2x
    s = im.astype(np.int32)
    mean4 = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(half, mean4)
    const = np.full((40, 50, 3), 173, np.uint8)
    assert (oracle_lib.resize_bilinear_u8(const, 123, 77) == 173).all()


def test_normalise_flip_pad_transpose_equal_numpy(oracle_lib):
    rs = np.random.RandomState(1)
    im = rs.randint(0, 256, (100, 160, 3)).astype(np.uint8)
    mean = np.array([123.675, 116.28, 103.53], np.float32)
    std = np.array([58.395, 57.12, 57.375], np.float32)
    for flip in (False, True):
        out, ishape, pshape, sf = oracle_lib.image_transform(im, (160, 100), flip, True, mean, std,
                                                             True, 32)
        assert ishape == (100, 160, 3) and pshape == (128, 160, 3) and sf == 1.0
        x = im.astype(np.float32)[:, :, ::-1]                     # BGR -> RGB
        x = (x - mean) / std                                      # mmcv.imnormalize
        if flip:
            x = x[:, ::-1]                                        # mmcv.imflip
        want = np.zeros((3, 128, 160), np.float32)
        want[:, :100, :] = x.transpose(2, 0, 1)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
