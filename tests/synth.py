"""Deterministic synthetic head outputs shared by the golden generator, the CPU
tests and the GPU parity tests (SURVEY.md 8d "head-boundary sets").

numpy's legacy RandomState stream is frozen by numpy's compatibility policy, so
the same seed yields the same tensors in the build container and on the GPU
box; fixtures only store seeds + expected outputs + an input checksum.
"""
import zlib

import numpy as np

STRIDES = (8, 16, 32, 64, 128)
A, C = 9, 80

# name -> (cls mean, cls std, iou std, reg std)
SETS = {
    'A': (-6.0, 2.0, 1.5, 0.5),    # tie-free wide
    'B': (-3.0, 2.0, 1.5, 0.5),    # dense: NMS stress
    'C': (-9.0, 1.5, 1.5, 0.5),    # sparse, trained-like (+ planted clusters)
}


def level_shapes(pad_h, pad_w, strides=STRIDES):
    """Feature-map sizes of ResNet+FPN(P3..P7) for a padded input."""
    shapes = []
    h, w = pad_h, pad_w
    cur = 1
    for s in strides:
        while cur < s:
            h, w = (h + 1) // 2, (w + 1) // 2
            cur *= 2
        shapes.append((h, w))
    return shapes


def head_outputs(seed, batch, pad_h, pad_w, kind='A', strides=STRIDES):
    """-> cls[L], reg[L], iou[L]  float32 arrays (B, ch, H, W)."""
    mu, sd, isd, rsd = SETS[kind]
    rs = np.random.RandomState(seed)
    cls, reg, iou = [], [], []
    for (h, w) in level_shapes(pad_h, pad_w, strides):
        cls.append((rs.standard_normal((batch, A * C, h, w)) * sd + mu).astype(np.float32))
        reg.append((rs.standard_normal((batch, A * 4, h, w)) * rsd).astype(np.float32))
        iou.append((rs.standard_normal((batch, A, h, w)) * isd).astype(np.float32))
    if kind == 'C':
        # planted clusters: neighbouring anchors of one class share a high score,
        # so NMS has real work to do (overlapping, same class).
        for b in range(batch):
            for _ in range(50):
                l = rs.randint(0, 3)
                h, w = cls[l].shape[2:]
                y, x = rs.randint(1, h - 1), rs.randint(1, w - 1)
                c = rs.randint(0, C)
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        for a in range(A):
                            cls[l][b, a * C + c, y + dy, x + dx] = \
                                np.float32(rs.uniform(0.0, 4.0))
    return cls, reg, iou


def checksum(arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).view(np.uint8), c)
    return c & 0xffffffff


def img_meta(img_h, img_w, pad_h, pad_w, scale_factor=1.0):
    return dict(ori_shape=(img_h, img_w, 3), img_shape=(img_h, img_w, 3),
                pad_shape=(pad_h, pad_w, 3), scale_factor=scale_factor, flip=False)


def train_targets(seed, batch, pad_h, pad_w, max_gt=8):
    """random gt boxes / labels for the training-loss fixtures."""
    rs = np.random.RandomState(seed)
    gts, labels = [], []
    for _ in range(batch):
        g = rs.randint(1, max_gt + 1)
        cx = rs.uniform(0.15 * pad_w, 0.85 * pad_w, g)
        cy = rs.uniform(0.15 * pad_h, 0.85 * pad_h, g)
        bw = rs.uniform(16, 0.5 * pad_w, g)
        bh = rs.uniform(16, 0.5 * pad_h, g)
        b = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
        b[:, 0::2] = np.clip(b[:, 0::2], 0, pad_w - 1)
        b[:, 1::2] = np.clip(b[:, 1::2], 0, pad_h - 1)
        gts.append(b.astype(np.float32))
        labels.append(rs.randint(1, C + 1, g).astype(np.int64))
    return gts, labels
