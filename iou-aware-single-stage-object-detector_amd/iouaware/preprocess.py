"""`ImageTransform` of the reference (mmdet/datasets/transforms.py:13-50) on the device: resize
(keep-ratio or exact), BGR->RGB, normalise, flip, pad to a multiple, HWC->CHW in one HIP launch
(csrc/preproc.hip) -- SURVEY 8f.3.  Sizes and scale factors follow mmcv 0.2.x
(`imrescale` / `imresize` / `impad_to_multiple`, restated: mmcv is a third-party package the
reference imports, INSTALL.md).  Images come in as uint8 HWC BGR arrays / tensors, exactly what
`mmcv.imread` hands to the reference's transform."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .ops import _ptr, _stream


def rescale_size(h, w, scale, keep_ratio=True):
    """-> (new_h, new_w, scale_factor): python float (keep_ratio, mmcv.imrescale) or the fp32
    4-vector (w_scale, h_scale, w_scale, h_scale) of transforms.py:35-38."""
    if keep_ratio:
        if isinstance(scale, (int, float)):
            if scale <= 0:
                raise ValueError('Invalid scale {}, must be positive.'.format(scale))
            sf = scale
        else:
            max_long, max_short = max(scale), min(scale)
            sf = min(max_long / max(h, w), max_short / min(h, w))
        return int(h * float(sf) + 0.5), int(w * float(sf) + 0.5), sf
    nw, nh = scale
    return int(nh), int(nw), np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32)


def _as_device_u8(img, device):
    if isinstance(img, np.ndarray):
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise TypeError('image must be uint8 (h, w, 3), got %s %s' % (img.dtype, img.shape))
        img = torch.from_numpy(np.ascontiguousarray(img))
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise TypeError('image must be uint8 (h, w, 3)')
    if not img.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.IouAwareLibraryError('ImageTransform needs a ROCm device: no CPU path')
        img = img.to(device if device is not None else 'cuda', non_blocking=True)
    return img.contiguous()


class ImageTransform(object):
    """Same constructor and call signature as the reference class; returns a device tensor."""

    def __init__(self, mean=(0, 0, 0), std=(1, 1, 1), to_rgb=True, size_divisor=None):
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb
        self.size_divisor = size_divisor

    def _pad(self, h, w):
        d = self.size_divisor
        if d is None:
            return h, w
        return int(np.ceil(h / d)) * d, int(np.ceil(w / d)) * d

    def __call__(self, img, scale, flip=False, keep_ratio=True, device=None):
        """-> (img (3, pad_h, pad_w) fp32 device tensor, img_shape, pad_shape, scale_factor)"""
        out, metas = self.batch([img], scale, [flip], keep_ratio, device=device)
        m = metas[0]
        ph, pw = m['pad_shape'][:2]
        return out[0, :, :ph, :pw], m['img_shape'], m['pad_shape'], m['scale_factor']

    def batch(self, imgs, scale, flips=None, keep_ratio=True, device=None, channels_last=False):
        """Whole batch in one launch.  -> (B,3,PH,PW) fp32 (PH, PW = the largest padded size of
        the batch, zero-filled like mmcv's collate), list of img_meta dicts (ori_shape, img_shape,
        pad_shape, scale_factor, flip -- mmdet/datasets/custom.py:300-305)."""
        B = len(imgs)
        flips = [False] * B if flips is None else list(flips)
        dev_imgs = [_as_device_u8(im, device) for im in imgs]
        dev = dev_imgs[0].device
        descs = (_lib.ImageDesc * B)()
        metas, PH, PW = [], 0, 0
        for b, im in enumerate(dev_imgs):
            h, w = int(im.shape[0]), int(im.shape[1])
            nh, nw, sf = rescale_size(h, w, scale, keep_ratio)
            if nh < 1 or nw < 1:
                # e.g. an 842 x 1 image under keep_ratio: int(1 * 0.27 + 0.5) = 0 columns.  The reference
                # fails here as well (mmcv.imrescale -> cv2.resize asserts a non-empty dsize,
                # transforms.py:35); a named error instead of the C-ABI's IA_E_ARG
                raise ValueError('image %d: %d x %d rescaled to an empty %d x %d image (scale %s, keep_ratio=%s)'
                                 % (b, h, w, nh, nw, scale, keep_ratio))
            ph, pw = self._pad(nh, nw)
            PH, PW = max(PH, ph), max(PW, pw)
            descs[b] = _lib.ImageDesc(im.data_ptr(), h, w, nh, nw, int(bool(flips[b])))
            metas.append(dict(ori_shape=(h, w, 3), img_shape=(nh, nw, 3), pad_shape=(ph, pw, 3),
                              scale_factor=sf, flip=bool(flips[b])))
        if channels_last:
            out = torch.empty((B, 3, PH, PW), dtype=torch.float32, device=dev,
                              memory_format=torch.channels_last)
        else:
            out = torch.empty((B, 3, PH, PW), dtype=torch.float32, device=dev)
        f3 = C.c_float * 3
        _lib.check(_lib.lib().ia_image_transform(descs, B, f3(*self.mean.tolist()),
                                                 f3(*self.std.tolist()), int(bool(self.to_rgb)), PH,
                                                 PW, int(bool(channels_last)), _ptr(out), _stream()),
                   'ia_image_transform')
        return out, metas
