#!/usr/bin/env python
"""the strided convolutions of the R-50 step at batch 8: im2col + GEMM / strided-batched GEMM
(frozen table, heuristic first result, all candidates timed) against the library convolution"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import bench  # noqa: E402
from iouaware import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'frozen'
ops.gemm_tuning(mode)
torch.backends.cudnn.benchmark = 'find' in sys.argv


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = int(os.environ.get('B', 8))
for C, H, W, n in ((128, 200, 336, 128), (256, 100, 168, 256), (512, 50, 84, 512), (2048, 25, 42, 256), (256, 13, 21, 256)):
    x = torch.randn(B, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(n, C, 3, 3, device='cuda').contiguous(memory_format=torch.channels_last) * 0.01
    wk = ops.conv3x3_weight_kn(w)
    b = torch.randn(n, device='cuda')
    t_own = timeit(lambda: ops.conv3x3_im2col(x, wk, b, 2, True))
    L = ops._lib.lib()
    col = ops._col_buffer(x.device, L.ia_im2col3x3_bytes(B, H, W, C, 2, 0))
    t_col = timeit(lambda: L.ia_im2col3x3_nhwc(ops._ptr(x), ops._ptr(col), B, H, W, C, 2, 0, ops._stream()))
    t_lib = timeit(lambda: ops.channel_affine_act_(F.conv2d(x, w, None, 2, 1), None, b, relu=True))
    gf = 2.0 * B * ((H + 1) // 2) * ((W + 1) // 2) * 9 * C * n / 1e9
    print('3x3/2 %4d -> %4d @ %3dx%3d: own %7.1f us (im2col %6.1f us, %.2f TB/s; GEMM %.0f TFLOP/s) | library conv + epilogue %7.1f us'
          % (C, n, H, W, t_own, t_col, (x.numel() * 4 * (1 + 2.25)) / t_col / 1e6, gf / (t_own - t_col) * 1e3 / 1e3, t_lib), flush=True)
for k, H, W, n in ((256, 200, 336, 512), (512, 100, 168, 1024), (1024, 50, 84, 2048)):
    x = torch.randn(B, k, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(n, k, 1, 1, device='cuda') * 0.01
    wk = w.view(n, k).t().contiguous()
    wc = w.contiguous(memory_format=torch.channels_last)
    t_own = timeit(lambda: ops.conv1x1_strided(x, wk, None, None, 2))
    t_lib = timeit(lambda: F.conv2d(x, wc, None, 2, 0))
    gf = 2.0 * B * (H // 2) * (W // 2) * k * n / 1e9
    print('1x1/2 %4d -> %4d @ %3dx%3d: own %7.1f us (%.0f TFLOP/s) | library conv %7.1f us' % (k, n, H, W, t_own, gf / t_own * 1e3, t_lib), flush=True)
x = torch.randn(B, 3, 800, 1344, device='cuda').contiguous(memory_format=torch.channels_last)
w = torch.randn(64, 3, 7, 7, device='cuda').contiguous(memory_format=torch.channels_last)
print('stem 7x7/2: library conv %.1f us' % timeit(lambda: F.conv2d(x, w, None, 2, 3)))
print(ops.gemm_table_stats())
ops.gemm_table_save(os.path.join(ROOT, 'gpurun_out', 'strided_%s.json' % mode), merge=False)
