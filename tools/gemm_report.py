"""Every library GEMM of one inference step (bench config 2: R-50, batch 8, 800 x 1344) with its shape,
its duration (HIP events around the call) and the fp32 MFMA rate it reaches: which products sit below
the ~125 TFLOP/s the large ones reach, and what they would gain."""
import os
import sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import torch
import bench
from iouaware import ops, winograd as wg

rec = []


def timed(name, shape_of, flops_of, fn):
    def wrapper(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        rec.append((name, shape_of(*a, **k), flops_of(*a, **k), e0, e1))
        return out
    return wrapper


def px(x):
    return x.shape[0] * x.shape[2] * x.shape[3]


ops.linear_bias_act = timed('1x1', lambda x, w, *a, **k: (px(x), w.shape[0], w.shape[1], 'res' if k.get('residual') is not None else ''),
                            lambda x, w, *a, **k: 2 * px(x) * w.shape[0] * w.shape[1], ops.linear_bias_act)
wg.batched_gemm = timed('wino-gemm', lambda v, u, out: (v.shape[0], v.shape[1], v.shape[2], u.shape[2]),
                        lambda v, u, out: 2 * v.shape[0] * v.shape[1] * v.shape[2] * u.shape[2], wg.batched_gemm)
_strided = ops.conv1x1_strided
ops.conv1x1_strided = timed('1x1-strided', lambda x, w, *a, **k: (px(x) // k.get('stride', 2) ** 2, w.shape[0], w.shape[1]),
                            lambda x, w, *a, **k: 2 * (px(x) // k.get('stride', 2) ** 2) * w.shape[0] * w.shape[1], _strided)
_im2col = ops.conv3x3_im2col
ops.conv3x3_im2col = timed('3x3-im2col', lambda x, w, *a, **k: (px(x) // k.get('stride', 2) ** 2, w.shape[0], w.shape[1]),
                           lambda x, w, *a, **k: 2 * (px(x) // k.get('stride', 2) ** 2) * w.shape[0] * w.shape[1], _im2col)
ops.conv1x1_chain = timed('1x1-chain', lambda x, w, b, r, w2, b2: (px(x), 64, 256, 64),
                          lambda x, w, b, r, w2, b2: 2 * px(x) * (64 * 256 + 256 * 64), ops.conv1x1_chain)

dev = torch.device('cuda', 0)
model = bench.build_model(dev, fuse=True, channels_last=True)
x = torch.randn(8, 3, bench.PAD_H, bench.PAD_W, device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3):
        model.forward_head(x)
    torch.cuda.synchronize()
    del rec[:]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model.forward_head(x)
    e1.record()
    torch.cuda.synchronize()
tot = e0.elapsed_time(e1)
rows = {}
for name, shape, fl, a, b in rec:
    ms = a.elapsed_time(b)
    k = (name, shape)
    r = rows.setdefault(k, [0, 0.0, fl])
    r[0] += 1
    r[1] += ms
print('forward_head with per-call events: %.2f ms; %d calls' % (tot, len(rec)))
print('%-12s %-34s %5s %9s %9s %8s' % ('kind', 'shape', 'calls', 'us/call', 'ms total', 'TFLOP/s'))
tt = tf = 0.0
for (name, shape), (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print('%-12s %-34s %5d %9.1f %9.3f %8.1f' % (name, str(shape), n, ms / n * 1e3, ms, fl * n / ms / 1e9))
    tt += ms
    tf += fl * n
print('all: %.2f ms, %.1f GFLOP, %.1f TFLOP/s average' % (tt, tf / 1e9, tf / tt / 1e9))
