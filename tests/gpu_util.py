"""helpers for the -m gpu parity tests"""
import numpy as np
import torch

import synth


def product_base_anchors(strides=synth.STRIDES, octave_base_scale=4, scales_per_octave=3,
                         ratios=(0.5, 1.0, 2.0)):
    """the base anchors the PRODUCT generates (iouaware/anchors.py, the generator behind
    head.geometry) -- the HIP path must not be fed the checker's anchors"""
    from iouaware.anchors import AnchorGenerator
    scales = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)]) \
        * octave_base_scale
    return np.stack([AnchorGenerator(s, scales, list(ratios)).base_anchors.numpy()
                     for s in strides])


def geometry(pad_h, pad_w, nms_pre, means=(0, 0, 0, 0), stds=(1, 1, 1, 1), softmax=False):
    """-> (HIP geometry built from the product's anchor generator, the ORACLE's base anchors for
    the checker side); the two generators must agree bit for bit"""
    import oracle
    from iouaware import ops
    sizes = synth.level_shapes(pad_h, pad_w)
    base_oracle = oracle.head_base_anchors(synth.STRIDES)
    base = product_base_anchors()
    assert base.dtype == np.float32 and np.array_equal(base, base_oracle)
    return ops.HeadGeometry(sizes, synth.STRIDES, base, synth.C, nms_pre=nms_pre, means=means,
                            stds=stds, softmax=softmax), base_oracle


def to_dev(arrs, dtype=torch.float32):
    return [torch.from_numpy(a).cuda().to(dtype) for a in arrs]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_bits(a, b):
    """bitwise equality, except that +0 == -0"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(((bits(a) == bits(b)) | ((a == 0) & (b == 0))).all())


def close(a, b, tol=1e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return True
    return bool((np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))).all())


def bf16_round(arrs):
    """round fp32 numpy arrays to bf16 (RNE) and return them as fp32"""
    return [torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy() for a in arrs]
