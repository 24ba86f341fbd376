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


def head_outputs_softmax(seed, batch, pad_h, pad_w, strides=STRIDES, nms_pre=300, margin=4e-5):
    """head outputs of a use_sigmoid_cls=False head (iou_aware_retina_head.py:506-507): the class
    tensor carries A * (C + 1) channels, channel 0 of an anchor = background (a +3 logit bias,
    like a trained softmax head: most anchors are background).  -> cls[L], reg[L], iou[L]

    VERDICT r5 item 9: the top-k of a level is decided by the row scores sqrt(max_fg softmax * sigmoid(iou)),
    and a softmax denominator summed in another order moves a score by a few ulp -- with independent random
    logits the closest pair among the top nms_pre + 1 scores of a level is ~1e-6 apart, i.e. "bit-exact
    indices" held by a handful of ulp.  So the generator SEPARATES them: per (image, level with more than
    nms_pre anchors), going down the nms_pre + 64 best rows (evaluated in fp64), every row that is less than
    `margin` (relative) below its predecessor gets its IoU logit lowered until it is -- a deterministic function
    of the seed (numpy only), so the GPU box regenerates the same inputs."""
    rs = np.random.RandomState(seed)
    cls, reg, iou = [], [], []
    for (h, w) in level_shapes(pad_h, pad_w, strides):
        c = (rs.standard_normal((batch, A, C + 1, h, w)) * 2.5).astype(np.float32)
        c[:, :, 0] += np.float32(3.0)
        r = (rs.standard_normal((batch, A * 4, h, w)) * 0.5).astype(np.float32)
        i = (rs.standard_normal((batch, A, h, w)) * 1.5).astype(np.float32)
        if nms_pre > 0 and A * h * w > nms_pre:
            c64 = c.astype(np.float64)
            e = np.exp(c64 - c64.max(2, keepdims=True))
            pmax = (e[:, :, 1:] / e.sum(2, keepdims=True)).max(2)              # (B, A, h, w)
            for b in range(batch):
                # reference row order: position-major, anchor-minor (cls_score.permute(1, 2, 0).reshape(-1, C + 1))
                pm = pmax[b].transpose(1, 2, 0).reshape(-1)
                il = i[b].transpose(1, 2, 0).reshape(-1).astype(np.float64)     # a copy: written back below
                sc = np.sqrt(pm / (1.0 + np.exp(-il)))
                order = np.argsort(-sc, kind='stable')[:nms_pre + 64]
                prev = None
                for row in order:
                    if prev is not None and sc[row] > prev * (1.0 - margin):
                        target = prev * (1.0 - 1.5 * margin)
                        q = target * target / pm[row]                          # sigmoid(iou') wanted
                        il[row] = np.log(q / (1.0 - q))
                        il[row] = np.float64(np.float32(il[row]))
                        sc[row] = np.sqrt(pm[row] / (1.0 + np.exp(-il[row])))
                    prev = sc[row]
                i[b] = il.reshape(h, w, A).transpose(2, 0, 1).astype(np.float32)
        cls.append(np.ascontiguousarray(c.reshape(batch, A * (C + 1), h, w)))
        reg.append(r)
        iou.append(i)
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


# ------------------------------------------------------------------ end-to-end fixtures
# "Trained-like" deterministic weights for the image -> detections fixtures
# (tests/golden/e2e_*.npz).  The reference's own random init gives degenerate, heavily tied
# scores; these weights keep activations at unit scale through the network and
# give head logits with the spread of set 'A' (tie-free, a few hundred boxes per class), so
# that a 1e-5 difference between two convolution algorithms cannot flip a selection.  The
# scheme depends on parameter NAMES and SHAPES only, which are identical in the reference and
# in this build (tests/golden/state_dict_keys.json), so both sides get bit-identical weights.
E2E_HEAD_GAIN = dict(retina_cls=1.4, retina_reg=0.45, retina_iou=1.6)   # -> std 2 / 0.5 / 1.5 on P3
E2E_CLS_BIAS = -6.0
E2E_LATERAL_GAIN = 0.125


def e2e_fill_state(state, seed):
    """fill a detector state dict (name -> torch tensor, modified in place) from `seed`."""
    import torch
    rs = np.random.RandomState(seed)
    for key in sorted(state.keys()):
        t = state[key]
        if key.endswith('num_batches_tracked'):
            continue
        shape = tuple(t.shape)
        leaf = key.split('.')[-1]
        if key.endswith('running_mean'):
            v = rs.standard_normal(shape) * 0.05
        elif key.endswith('running_var'):
            v = rs.uniform(0.8, 1.2, shape)
        elif t.dim() == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = np.sqrt(2.0 / fan_in)                       # keeps the scale through a ReLU
            if key.startswith('neck.'):
                gain = np.sqrt(1.0 / fan_in)                   # FPN convs have no activation
                if '.lateral_convs.' in key or '.fpn_convs.3.' in key:
                    gain *= E2E_LATERAL_GAIN                   # brings C3..C5 (P6 reads C5) back to unit scale
            for name, g in E2E_HEAD_GAIN.items():
                if ('.%s.' % name) in key:
                    gain = g * np.sqrt(1.0 / fan_in)
            v = rs.standard_normal(shape) * gain
        elif key.startswith('backbone.') and leaf == 'weight':   # BatchNorm scale
            v = rs.uniform(0.8, 1.2, shape)
            if '.bn3.' in key:
                v = v * 0.25                                   # residual branch: no blow-up
        elif leaf == 'bias':
            v = rs.standard_normal(shape) * (0.05 if key.startswith('backbone.') else 0.02)
            if '.retina_cls.' in key:
                v = E2E_CLS_BIAS + rs.standard_normal(shape) * 0.5
        else:
            raise KeyError('e2e_fill_state: no rule for %s %s' % (key, shape))
        t.copy_(torch.from_numpy(np.asarray(v, np.float32)))


def e2e_image(seed, batch, pad_h, pad_w, img_h, img_w):
    """N(0,1) image, zero in the padding like ImageTransform's pad (transforms.py:44-46)."""
    rs = np.random.RandomState(seed)
    img = rs.standard_normal((batch, 3, pad_h, pad_w)).astype(np.float32)
    img[:, :, img_h:, :] = 0
    img[:, :, :, img_w:] = 0
    return img


def e2e_gts(seed, img_h, img_w):
    """gt boxes / labels for the test-time call (dead inputs in the reference, :517-524)."""
    g, l = train_targets(seed, 1, img_h, img_w, max_gt=4)
    return g[0], l[0]


def mnms_big_inputs(seed=23, n=12000):
    """boxes (n, 4) and scores (n, 4: background + 3 classes) of tests/golden/mnms_big.npz:
    more boxes than one batched launch holds, thousands of NMS survivors, tie-free scores"""
    rs = np.random.RandomState(seed)
    xy = rs.uniform(0, 3000, (n, 2))
    boxes = np.concatenate([xy, xy + rs.uniform(8, 60, (n, 2))], 1).astype(np.float32)
    sc = np.zeros((n, 4), np.float32)
    perm = [rs.permutation(3 * n)[:n] for _ in range(3)]
    for c in range(3):
        sc[:, c + 1] = ((perm[c].astype(np.float64) * 3 + c + 0.5) / (9.0 * n + 3)).astype(np.float32)
    assert np.unique(sc[:, 1:]).size == 3 * n
    return boxes, sc
