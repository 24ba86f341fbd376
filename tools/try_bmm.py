"""Scratch: fp32 batched GEMM throughput (rocBLAS / hipBLASLt through torch) at the shapes a
Winograd F(4x4,3x3) head would produce: 36 x [tiles x Cin] x [Cin x Cout]."""
import torch, time
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n
for (bt, M, K, N) in [(36, 11456, 256, 256), (36, 8400, 256, 256), (36, 11456, 256, 768), (16, 45824, 256, 256),
                      (36, 2184, 256, 256), (1, 36 * 11456, 256, 256), (36, 11456, 256, 720), (36, 11456, 256, 64)]:
    a = torch.randn(bt, M, K, device='cuda'); b = torch.randn(bt, K, N, device='cuda')
    out = torch.empty(bt, M, N, device='cuda')
    t = bench(lambda: torch.bmm(a, b, out=out))
    print('bmm %2d x [%6d x %3d] x [%3d x %3d]: %.3f ms  %.1f TFLOP/s' % (bt, M, K, K, N, t * 1e3, 2 * bt * M * K * N / t / 1e12))
    at = a.transpose(1, 2).contiguous()   # K-major A (tiles fastest)
    t = bench(lambda: torch.bmm(at.transpose(1, 2), b, out=out))
    print('   A stored [K x M]: %.3f ms  %.1f TFLOP/s' % (t * 1e3, 2 * bt * M * K * N / t / 1e12))
