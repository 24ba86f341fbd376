// Micro-benchmark (not product code): access-pattern variants for the row-max read of one
// level, (B, A*C, HW) fp32, to find what limits k_rowmax.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int A = 9, C = 80;

// ceiling: plain streaming max-reduce over the whole buffer
__global__ void k_stream(const float4 *p, size_t n4, float *out)
{
    float m = -1e30f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i];
        m = fmaxf(fmaxf(m, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    if (m == 12345.f) out[0] = m;
}

// V0: block (64, A): wave = anchor, 256 positions/tile, loop classes (current product kernel)
template <int UNROLL>
__global__ void __launch_bounds__(1024) k_v0(const float *cls, int HW, int tiles, float *out)
{
    const int lane = threadIdx.x, an = threadIdx.y;
    const int b = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int pp = t * 256 + lane * 4;
    if (pp >= HW) return;
    const float *src = cls + ((size_t)b * A + an) * C * HW + pp;
    float m0 = -1e30f, m1 = m0, m2 = m0, m3 = m0;
#pragma unroll UNROLL
    for (int c = 0; c < C; ++c) {
        float4 v = *reinterpret_cast<const float4 *>(src + (size_t)c * HW);
        m0 = fmaxf(m0, v.x); m1 = fmaxf(m1, v.y); m2 = fmaxf(m2, v.z); m3 = fmaxf(m3, v.w);
    }
    float *o = out + ((size_t)b * HW + pp) * A + an;
    o[0] = m0; o[A] = m1; o[2 * A] = m2; o[3 * A] = m3;
}

// V1: block = W waves on W adjacent 256-position chunks of the SAME plane; one anchor per block
template <int W>
__global__ void __launch_bounds__(64 * W) k_v1(const float *cls, int HW, int tiles, float *out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int bid = blockIdx.x;
    const int t = bid % tiles; bid /= tiles;
    const int an = bid % A; const int b = bid / A;
    const int pp = (t * W + wv) * 256 + lane * 4;
    if (pp >= HW) return;
    const float *src = cls + ((size_t)b * A + an) * C * HW + pp;
    float m0 = -1e30f, m1 = m0, m2 = m0, m3 = m0;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        float4 v = *reinterpret_cast<const float4 *>(src + (size_t)c * HW);
        m0 = fmaxf(m0, v.x); m1 = fmaxf(m1, v.y); m2 = fmaxf(m2, v.z); m3 = fmaxf(m3, v.w);
    }
    float *o = out + ((size_t)b * HW + pp) * A + an;
    o[0] = m0; o[A] = m1; o[2 * A] = m2; o[3 * A] = m3;
}

// V2: like V0 but each lane owns 8 consecutive positions (2 x float4 per class, 2 KiB runs per wave)
__global__ void __launch_bounds__(1024) k_v2(const float *cls, int HW, int tiles, float *out)
{
    const int lane = threadIdx.x, an = threadIdx.y;
    const int b = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int pp = t * 512 + lane * 4;
    const float *src = cls + ((size_t)b * A + an) * C * HW + pp;
    const bool ok0 = pp < HW, ok1 = pp + 256 < HW;
    float m[8];
    for (int j = 0; j < 8; ++j) m[j] = -1e30f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        if (ok0) { float4 v = *reinterpret_cast<const float4 *>(src + (size_t)c * HW);
            m[0] = fmaxf(m[0], v.x); m[1] = fmaxf(m[1], v.y); m[2] = fmaxf(m[2], v.z); m[3] = fmaxf(m[3], v.w); }
        if (ok1) { float4 v = *reinterpret_cast<const float4 *>(src + (size_t)c * HW + 256);
            m[4] = fmaxf(m[4], v.x); m[5] = fmaxf(m[5], v.y); m[6] = fmaxf(m[6], v.z); m[7] = fmaxf(m[7], v.w); }
    }
    float *o = out + ((size_t)b * HW + pp) * A + an;
    if (ok0) { o[0] = m[0]; o[A] = m[1]; o[2 * A] = m[2]; o[3 * A] = m[3]; }
    if (ok1) { o += 256 * A; o[0] = m[4]; o[A] = m[5]; o[2 * A] = m[6]; o[3 * A] = m[7]; }
}

// V3: class-split: block (64, A, S): S waves per anchor each take C/S classes; combine in LDS
template <int S>
__global__ void __launch_bounds__(1024) k_v3(const float *cls, int HW, int tiles, float *out)
{
    __shared__ float red[S][A][256];
    const int lane = threadIdx.x, an = threadIdx.y, sp = threadIdx.z;
    const int b = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const int pp = t * 256 + lane * 4;
    const bool ok = pp < HW;
    const float *src = cls + ((size_t)b * A + an) * C * HW + pp;
    float m0 = -1e30f, m1 = m0, m2 = m0, m3 = m0;
    if (ok) {
#pragma unroll 10
        for (int c = sp * (C / S); c < (sp + 1) * (C / S); ++c) {
            float4 v = *reinterpret_cast<const float4 *>(src + (size_t)c * HW);
            m0 = fmaxf(m0, v.x); m1 = fmaxf(m1, v.y); m2 = fmaxf(m2, v.z); m3 = fmaxf(m3, v.w);
        }
    }
    red[sp][an][lane * 4] = m0; red[sp][an][lane * 4 + 1] = m1;
    red[sp][an][lane * 4 + 2] = m2; red[sp][an][lane * 4 + 3] = m3;
    __syncthreads();
    if (sp == 0 && ok) {
        for (int s = 1; s < S; ++s) {
            m0 = fmaxf(m0, red[s][an][lane * 4]); m1 = fmaxf(m1, red[s][an][lane * 4 + 1]);
            m2 = fmaxf(m2, red[s][an][lane * 4 + 2]); m3 = fmaxf(m3, red[s][an][lane * 4 + 3]);
        }
        float *o = out + ((size_t)b * HW + pp) * A + an;
        o[0] = m0; o[A] = m1; o[2 * A] = m2; o[3 * A] = m3;
    }
}


typedef float f4 __attribute__((ext_vector_type(4)));
// V4: single-wave blocks, non-temporal loads
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(64) k_v4(const float *cls, int HW, int tiles, float *out)
{
    const int lane = threadIdx.x;
    int bid = blockIdx.x;
    const int t = bid % tiles; bid /= tiles;
    const int an = bid % A; const int b = bid / A;
    int pp = t * 256 + lane * 4;
    const bool ok = pp < HW;
    if (!ok) pp = HW - 4;
    const float *src = cls + ((size_t)b * A + an) * C * HW + pp;
    float m0 = -1e30f, m1 = m0, m2 = m0, m3 = m0;
#pragma unroll UNROLL
    for (int c = 0; c < C; ++c) {
        const f4 *q = reinterpret_cast<const f4 *>(src + (size_t)c * HW);
        f4 v = NT ? __builtin_nontemporal_load(q) : *q;
        m0 = fmaxf(m0, v.x); m1 = fmaxf(m1, v.y); m2 = fmaxf(m2, v.z); m3 = fmaxf(m3, v.w);
    }
    if (ok) { float *o = out + ((size_t)b * HW + pp) * A + an; o[0] = m0; o[A] = m1; o[2 * A] = m2; o[3 * A] = m3; }
}
// V5: single-wave blocks, 8 positions per lane (two float4 per class, 2 KiB per wave per plane)
template <bool NT>
__global__ void __launch_bounds__(64) k_v5(const float *cls, int HW, int tiles, float *out)
{
    const int lane = threadIdx.x;
    int bid = blockIdx.x;
    const int t = bid % tiles; bid /= tiles;
    const int an = bid % A; const int b = bid / A;
    int p0 = t * 512 + lane * 4, p1 = p0 + 256;
    const bool ok0 = p0 < HW, ok1 = p1 < HW;
    if (!ok0) p0 = HW - 4;
    if (!ok1) p1 = HW - 4;
    const float *base = cls + ((size_t)b * A + an) * C * HW;
    float m[8];
    for (int j = 0; j < 8; ++j) m[j] = -1e30f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const f4 *q0 = reinterpret_cast<const f4 *>(base + (size_t)c * HW + p0);
        const f4 *q1 = reinterpret_cast<const f4 *>(base + (size_t)c * HW + p1);
        f4 v = NT ? __builtin_nontemporal_load(q0) : *q0;
        f4 w = NT ? __builtin_nontemporal_load(q1) : *q1;
        m[0] = fmaxf(m[0], v.x); m[1] = fmaxf(m[1], v.y); m[2] = fmaxf(m[2], v.z); m[3] = fmaxf(m[3], v.w);
        m[4] = fmaxf(m[4], w.x); m[5] = fmaxf(m[5], w.y); m[6] = fmaxf(m[6], w.z); m[7] = fmaxf(m[7], w.w);
    }
    if (ok0) { float *o = out + ((size_t)b * HW + p0) * A + an; o[0] = m[0]; o[A] = m[1]; o[2 * A] = m[2]; o[3 * A] = m[3]; }
    if (ok1) { float *o = out + ((size_t)b * HW + p1) * A + an; o[0] = m[4]; o[A] = m[5]; o[2 * A] = m[6]; o[3 * A] = m[7]; }
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    const int HW = argc > 2 ? atoi(argv[2]) : 16800;
    const size_t n = (size_t)B * A * C * HW;
    float *cls, *out;
    CK(hipMalloc(&cls, n * 4));
    CK(hipMalloc(&out, (size_t)B * HW * A * 4 + 4096));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 6.f;
    CK(hipMemcpy(cls, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int it = 20;
        for (int i = 0; i < it; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        printf("%-28s %8.3f us  %7.1f GB/s\n", name, ms * 1e3, n * 4 / ms / 1e6);
    };
    printf("B=%d HW=%d bytes=%.1f MB\n", B, HW, n * 4 / 1e6);
    timeit("stream ceiling 2048x256", [&] { k_stream<<<2048, 256>>>((const float4 *)cls, n / 4, out); });
    timeit("stream ceiling 8192x256", [&] { k_stream<<<8192, 256>>>((const float4 *)cls, n / 4, out); });
    { int tiles = (HW + 255) / 256;
      timeit("v0 (64,9) tile256 u8", [&] { k_v0<8><<<B * tiles, dim3(64, A)>>>(cls, HW, tiles, out); });
      timeit("v0 (64,9) tile256 u16", [&] { k_v0<16><<<B * tiles, dim3(64, A)>>>(cls, HW, tiles, out); });
      timeit("v0 (64,9) tile256 u4", [&] { k_v0<4><<<B * tiles, dim3(64, A)>>>(cls, HW, tiles, out); });
      timeit("v4 1wave u8", [&] { k_v4<8, false><<<B * A * tiles, 64>>>(cls, HW, tiles, out); });
      timeit("v4 1wave u8 nt", [&] { k_v4<8, true><<<B * A * tiles, 64>>>(cls, HW, tiles, out); });
      timeit("v4 1wave u16", [&] { k_v4<16, false><<<B * A * tiles, 64>>>(cls, HW, tiles, out); });
      timeit("v4 1wave u16 nt", [&] { k_v4<16, true><<<B * A * tiles, 64>>>(cls, HW, tiles, out); });
      timeit("v4 1wave u4 nt", [&] { k_v4<4, true><<<B * A * tiles, 64>>>(cls, HW, tiles, out); }); }
    { int tiles = (HW + 511) / 512;
      timeit("v5 1wave tile512", [&] { k_v5<false><<<B * A * tiles, 64>>>(cls, HW, tiles, out); });
      timeit("v5 1wave tile512 nt", [&] { k_v5<true><<<B * A * tiles, 64>>>(cls, HW, tiles, out); }); }
    { int tiles = (HW + 1023) / 1024;
      timeit("v1 4 waves/plane-run", [&] { k_v1<4><<<B * A * tiles, 256>>>(cls, HW, tiles, out); }); }
    { int tiles = (HW + 2047) / 2048;
      timeit("v1 8 waves/plane-run", [&] { k_v1<8><<<B * A * tiles, 512>>>(cls, HW, tiles, out); }); }
    { int tiles = (HW + 255) / 256;
      timeit("v1 1 wave (64-thr blocks)", [&] { k_v1<1><<<B * A * tiles, 64>>>(cls, HW, tiles, out); }); }
    { int tiles = (HW + 511) / 512;
      timeit("v2 (64,9) tile512", [&] { k_v2<<<B * tiles, dim3(64, A)>>>(cls, HW, tiles, out); }); }
    return 0;
}
