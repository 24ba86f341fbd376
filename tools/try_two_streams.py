"""Scratch: the inference step as two half-batches on two streams (GEMM of one overlapping the
HBM-bound transforms of the other) against one batch-8 forward"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import torch, bench
from iouaware import ops
torch.backends.cudnn.benchmark = True
ops.gemm_tuning('all')
dev = torch.device('cuda', 0)
model = bench.build_model(dev, channels_last=True)
B = bench.BATCH
x = torch.randn(B, 3, bench.PAD_H, bench.PAD_W, device=dev).contiguous(memory_format=torch.channels_last)
metas = bench.metas(B)
def one():
    return model.simple_test_device(x, metas, rescale=True)
def split(n, streams):
    h = B // n
    cur = torch.cuda.current_stream()
    outs = []
    for i, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(model.simple_test_device(x[i * h:(i + 1) * h], metas[i * h:(i + 1) * h], rescale=True))
    for s in streams:
        cur.wait_stream(s)
    return outs
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    print('batch 8, one stream      %.2f ms' % timeit(one))
    for n in (2, 4):
        streams = [torch.cuda.Stream() for _ in range(n)]
        print('%d x batch %d on %d streams %.2f ms' % (n, B // n, n, timeit(lambda: split(n, streams))))
    s1 = [torch.cuda.current_stream()] * 2
    print('2 x batch 4 on ONE stream %.2f ms' % timeit(lambda: split(2, s1)))
    a = one(); b = split(2, [torch.cuda.Stream() for _ in range(2)])
    torch.cuda.synchronize()
    print('identical detections', all(torch.equal(a[k], torch.cat([b[0][k], b[1][k]])) for k in range(4)))
