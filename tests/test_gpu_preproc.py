"""GPU (MI355X): the fused pre-processing kernel (csrc/preproc.hip) through the C-ABI against the
oracle -- bit for bit (uint8 interpolation is integer work, the normalisation is exact IEEE) --
for keep-ratio and exact resizes, up- and down-scaling, flips, padding, both output layouts and
mixed-size batches."""
import numpy as np
import pytest
import torch

import gpu_util as G

pytestmark = pytest.mark.gpu

NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


def _img(rs, h, w, smooth=False):
    if not smooth:
        return rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([(np.sin(xx / 37.0) + np.cos(yy / 23.0)) * 60 + 128, xx * 0.3 + yy * 0.1,
                     (xx + yy) % 200], 2).astype(np.uint8)


@pytest.mark.parametrize('h,w,scale,keep', [
    (480, 640, (1333, 800), True),        # COCO landscape -> 800x1067, pad 800x1088
    (640, 427, (1333, 800), True),        # portrait -> 1199x800
    (1200, 1800, (1333, 800), True),      # down-scaling
    (375, 500, (1000, 600), True),
    (100, 100, (200, 200), True),         # exactly 2x up
    (400, 600, (300, 200), True),         # exactly 2x down (cv2's INTER_AREA fast path: same bytes)
    (97, 131, (131, 97), True),           # same size -> copy
    (240, 320, (512, 384), False),        # imresize, anisotropic 4-vector scale factor
    (33, 47, (1333, 800), True),          # tiny source, heavy up-scaling
])
def test_image_transform_equals_oracle(oracle_lib, h, w, scale, keep):
    from iouaware.preprocess import ImageTransform
    rs = np.random.RandomState(h * 7 + w)
    tf = ImageTransform(size_divisor=32, **NORM)
    for smooth in (False, True):
        img = _img(rs, h, w, smooth)
        for flip in (False, True):
            want, ishape, pshape, sf = oracle_lib.image_transform(img, scale, flip, keep,
                                                                  size_divisor=32, **NORM)
            got, gi, gp, gsf = tf(img, scale, flip, keep)
            assert tuple(gi) == tuple(ishape) and tuple(gp) == tuple(pshape)
            assert np.array_equal(np.asarray(gsf, np.float64), np.asarray(sf, np.float64))
            assert got.is_cuda and got.dtype == torch.float32
            assert G.same_bits(got.cpu().numpy(), want), (h, w, flip, smooth)


def test_an_empty_rescaled_image_is_a_named_error():
    """the reference fails on such an input as well (cv2.resize asserts a non-empty dsize); the host side
    names the image instead of passing a 0-wide destination to the C-ABI (IA_E_ARG)"""
    from iouaware.preprocess import ImageTransform
    tf = ImageTransform(size_divisor=32, **NORM)
    with pytest.raises(ValueError, match='empty 227 x 0'):
        tf(np.zeros((842, 1, 3), np.uint8), (172, 227), False, True)
    with pytest.raises(ValueError, match='image 1'):
        tf.batch([np.zeros((64, 64, 3), np.uint8), np.zeros((842, 1, 3), np.uint8)], (172, 227))


def test_batch_mixed_sizes_and_channels_last(oracle_lib):
    from iouaware.preprocess import ImageTransform
    rs = np.random.RandomState(3)
    imgs = [_img(rs, 480, 640), _img(rs, 640, 480), _img(rs, 427, 640), _img(rs, 500, 375)] * 5
    flips = [bool(i % 2) for i in range(len(imgs))]                   # 20 images: two launches
    tf = ImageTransform(size_divisor=32, **NORM)
    out, metas = tf.batch(imgs, (1333, 800), flips)
    out_cl, _ = tf.batch(imgs, (1333, 800), flips, channels_last=True)
    assert out_cl.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out, out_cl.contiguous())
    PH, PW = out.shape[2:]
    assert (PH, PW) == (1088, 1216)             # 640x480 -> 1067x800 -> 1088; 427x640 -> 800x1199 -> 1216
    for b, (im, fl) in enumerate(zip(imgs, flips)):
        want, ishape, pshape, sf = oracle_lib.image_transform(im, (1333, 800), fl, True,
                                                              size_divisor=32, **NORM)
        assert metas[b]['img_shape'] == ishape and metas[b]['pad_shape'] == pshape
        assert metas[b]['scale_factor'] == sf and metas[b]['flip'] == fl
        assert metas[b]['ori_shape'] == im.shape
        ph, pw = pshape[:2]
        o = out[b].cpu().numpy()
        assert G.same_bits(o[:, :ph, :pw], want)
        assert not o[:, ph:, :].any() and not o[:, :, pw:].any()      # batch padding is zero


def test_transform_feeds_the_detector():
    """uint8 image -> ImageTransform -> detector forward on the device (no host round trip)"""
    import iouaware
    from iouaware.config import ConfigDict
    from test_host_model import R50_MODEL, TEST_CFG
    rs = np.random.RandomState(0)
    tf = iouaware.ImageTransform(size_divisor=32, **NORM)
    imgs, metas = tf.batch([_img(rs, 120, 160, True), _img(rs, 96, 160, True)], (256, 160))
    torch.manual_seed(0)
    m = iouaware.build_detector(ConfigDict(R50_MODEL), test_cfg=ConfigDict(TEST_CFG)).cuda().eval()
    with torch.no_grad():
        res = m.simple_test_batch(imgs, metas, rescale=True)
    assert len(res) == 2 and all(len(r) == 80 for r in res)       # bbox2result lists per image
    # serving loop: two batches in flight, collected out of step with their submission
    # (the library convolutions of the module path are not bit-reproducible from call to call --
    # head outputs move by an ulp, boxes by ~1e-5 -- hence a tolerance, not array_equal)
    same = lambda x, y: all(a.shape == b.shape and (a.size == 0 or np.abs(a - b).max() < 1e-4)
                            for ra, rb in zip(x, y) for a, b in zip(ra, rb))
    flipped = imgs.flip(0).contiguous()
    with torch.no_grad():
        res = m.simple_test_batch(imgs, metas, rescale=True)          # conv algorithms settled
        again = m.simple_test_batch(flipped, metas[::-1], rescale=True)
        p1 = m.simple_test_batch_submit(imgs, metas, rescale=True)
        p2 = m.simple_test_batch_submit(flipped, metas[::-1], rescale=True)
        res_after = m.simple_test_batch(imgs, metas, rescale=True)
    assert same(res_after, res)
    assert same(p2.collect(), again)
    assert same(p1.collect(), res)
