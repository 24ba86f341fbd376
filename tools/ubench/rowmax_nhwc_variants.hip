// Micro-benchmark (not product code): variants of the channels-last row-max read -- rows of
// C = 80 contiguous fp32 logits, one maximum per row -- to find what limits k_rowmax_nhwc.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/rowmax_nhwc_variants.hip -o /tmp/rmn && /tmp/rmn
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int C = 80, VPR = 20;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_stream(const float4 *p, size_t n4, float *out)
{
    float m = -1e30f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i];
        m = fmaxf(fmaxf(m, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    if (m == 12345.f) out[0] = m;
}

template <bool NT> __device__ __forceinline__ float ld_max(const float *p)
{
    f32x4 q = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p))
                 : *reinterpret_cast<const f32x4 *>(p);
    return fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w));
}

// N0: W waves per block, each wave: 64 rows, 20 coalesced loads/lane, LDS transpose, lane = row
template <int W, bool NT>
__global__ void __launch_bounds__(64 * W) k_n0(const float *cls, size_t rows, float *out)
{
    __shared__ float s[W][64 * (VPR + 1)];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t r0 = ((size_t)blockIdx.x * W + wv) * 64;
    if (r0 >= rows) return;
    const float *src = cls + r0 * C;
    float m[VPR];
#pragma unroll
    for (int k = 0; k < VPR; ++k) m[k] = ld_max<NT>(src + (size_t)(k * 64 + lane) * 4);
#pragma unroll
    for (int k = 0; k < VPR; ++k) {
        const int f = k * 64 + lane, row = f / VPR, c4 = f - row * VPR;
        s[wv][row * (VPR + 1) + c4] = m[k];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): LDS writes of this wave are done
    const float *sr = s[wv] + lane * (VPR + 1);
    float mx = sr[0];
#pragma unroll
    for (int c4 = 1; c4 < VPR; ++c4) mx = fmaxf(mx, sr[c4]);
    out[r0 + lane] = mx;
}

// N1: like N0 with a real __syncthreads (the product kernel today, W = 1)
template <bool NT>
__global__ void __launch_bounds__(64) k_n1(const float *cls, size_t rows, float *out)
{
    __shared__ float s[64 * 33];
    const int lane = threadIdx.x;
    const size_t r0 = (size_t)blockIdx.x * 64;
    const float *src = cls + r0 * C;
#pragma unroll
    for (int k = 0; k < VPR; ++k) {
        const int f = k * 64 + lane, row = f / VPR, c4 = f - row * VPR;
        s[row * (VPR + 1) + c4] = ld_max<NT>(src + (size_t)f * 4);
    }
    __syncthreads();
    const float *sr = s + lane * (VPR + 1);
    float mx = sr[0];
    for (int c4 = 1; c4 < VPR; ++c4) mx = fmaxf(mx, sr[c4]);
    out[r0 + lane] = mx;
}

// N2: thread = row, 20 strided 16-byte loads (each instruction touches 64 different rows)
template <bool NT>
__global__ void __launch_bounds__(64) k_n2(const float *cls, size_t rows, float *out)
{
    const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
    const float *src = cls + r * C;
    float mx = -1e30f;
#pragma unroll
    for (int k = 0; k < VPR; ++k) mx = fmaxf(mx, ld_max<NT>(src + k * 4));
    out[r] = mx;
}

// N3: 4 lanes per row, 5 consecutive vectors per lane, quad reduction by DPP -- no LDS
template <bool NT>
__global__ void __launch_bounds__(64) k_n3(const float *cls, size_t rows, float *out)
{
    const int lane = threadIdx.x;
    const size_t r0 = (size_t)blockIdx.x * 64;          // 64 rows per wave = 4 groups of 16 rows
    float res[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float *src = cls + (r0 + g * 16) * C + (size_t)lane * 20;   // lane's quarter row
        float m = -1e30f;
#pragma unroll
        for (int j = 0; j < 5; ++j) m = fmaxf(m, ld_max<NT>(src + j * 4));
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        res[g] = m;
    }
    if ((lane & 3) == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) out[r0 + g * 16 + (lane >> 2)] = res[g];
    }
}

// N4: N0 mapping, two chunks per wave with the second chunk's loads issued before the first
// chunk is reduced (40 loads in flight per lane)
template <bool NT>
__global__ void __launch_bounds__(64) k_n4(const float *cls, size_t rows, float *out)
{
    __shared__ float s[64 * (VPR + 1)];
    const int lane = threadIdx.x;
    const size_t r0 = (size_t)blockIdx.x * 128;
    float m0[VPR], m1[VPR];
#pragma unroll
    for (int k = 0; k < VPR; ++k) m0[k] = ld_max<NT>(cls + r0 * C + (size_t)(k * 64 + lane) * 4);
#pragma unroll
    for (int k = 0; k < VPR; ++k) m1[k] = ld_max<NT>(cls + (r0 + 64) * C + (size_t)(k * 64 + lane) * 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int k = 0; k < VPR; ++k) {
            const int f = k * 64 + lane, row = f / VPR, c4 = f - row * VPR;
            s[row * (VPR + 1) + c4] = h ? m1[k] : m0[k];
        }
        __syncthreads();
        const float *sr = s + lane * (VPR + 1);
        float mx = sr[0];
#pragma unroll
        for (int c4 = 1; c4 < VPR; ++c4) mx = fmaxf(mx, sr[c4]);
        out[r0 + h * 64 + lane] = mx;
        __syncthreads();
    }
}

int main(int argc, char **argv)
{
    const size_t rows = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)8 * 151200;   // P3, batch 8
    const size_t n = rows * C;
    float *cls, *out;
    CK(hipMalloc(&cls, n * 4 + 65536));
    CK(hipMalloc(&out, rows * 4 + 65536));
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
        printf("%-36s %8.3f us  %7.1f GB/s\n", name, ms * 1e3, n * 4 / ms / 1e6);
    };
    printf("rows=%zu bytes=%.1f MB\n", rows, n * 4 / 1e6);
    const unsigned g64 = (unsigned)(rows / 64);
    timeit("stream ceiling 8192x256", [&] { k_stream<<<8192, 256>>>((const float4 *)cls, n / 4, out); });
    timeit("n1 product (syncthreads) nt", [&] { k_n1<true><<<g64, 64>>>(cls, rows, out); });
    timeit("n1 product (syncthreads)", [&] { k_n1<false><<<g64, 64>>>(cls, rows, out); });
    timeit("n0 W=1 regs+wave_barrier nt", [&] { k_n0<1, true><<<g64, 64>>>(cls, rows, out); });
    timeit("n0 W=1 regs+wave_barrier", [&] { k_n0<1, false><<<g64, 64>>>(cls, rows, out); });
    timeit("n0 W=4 nt", [&] { k_n0<4, true><<<g64 / 4, 256>>>(cls, rows, out); });
    timeit("n0 W=4", [&] { k_n0<4, false><<<g64 / 4, 256>>>(cls, rows, out); });
    timeit("n2 thread=row nt", [&] { k_n2<true><<<g64, 64>>>(cls, rows, out); });
    timeit("n2 thread=row", [&] { k_n2<false><<<g64, 64>>>(cls, rows, out); });
    timeit("n3 quad/row no LDS nt", [&] { k_n3<true><<<g64, 64>>>(cls, rows, out); });
    timeit("n3 quad/row no LDS", [&] { k_n3<false><<<g64, 64>>>(cls, rows, out); });
    timeit("n4 2 chunks, 40 loads nt", [&] { k_n4<true><<<g64 / 2, 64>>>(cls, rows, out); });
    timeit("n4 2 chunks, 40 loads", [&] { k_n4<false><<<g64 / 2, 64>>>(cls, rows, out); });
    return 0;
}
