// 1x1 convolutions of ResNet stage 1 as streaming MFMA kernels (fp32): y = relu?(x . W + bias
// (+ residual)) for K x N = 64 x 256, 256 x 64, 64 x 64 (reference mmdet/models/backbones/
// resnet.py:215-255, the bottleneck's conv1 / conv3 with the folded BatchNorm).  These products
// have 7 % of the network's multiply-adds and are HBM-bound (a 64 -> 256 convolution with its
// residual moves 1.24 GB for 17.6 GFLOP); the library GEMM runs them at 3.8-4.0 TB/s.
//
// The whole weight matrix (K * N <= 16 384 floats = 64 KB) sits in LDS; a wavefront streams
// 16-pixel tiles with NO barrier after the weights are in: it computes D^T = W^T (N x K) . X^T on
// v_mfma_f32_16x16x4_f32, so that a lane ends up with FOUR CONSECUTIVE OUTPUT CHANNELS of one
// pixel -- residual loads, the accumulator initialisation and the stores are 16 bytes per lane,
// 64 contiguous bytes per pixel.  The activation rows are read with 16-byte loads as well: K is a
// reduction index, so WHICH k a lane group feeds into MFMA step s is free as long as the weight
// fragment uses the same one: lane (pixel p, group q) loads x[p][16 j + 4 q .. + 3] and step
// s = 4 j + c consumes k = 16 j + 4 q + c on both operands.
// All loads of a tile (activation + residual, 20 x 16 bytes per lane for 64 -> 256) are issued
// before the first MFMA; the residual lands directly in the accumulators.
#include <string.h>
#include <mutex>
#include <vector>
#include "ia_internal.hpp"

namespace ia {

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct Conv1Args {
    const float *x, *w, *bias, *res;      // x (P, K), w (K, N), bias (N) or NULL, res (P, N) or NULL
    float *y;                             // (P, N)
    int64_t P;
    int32_t relu, tiles;
    int64_t xs, ws, ys;                   // batched use (blockIdx.y = item): element strides of x / w / y
};

constexpr int kC1Threads = 512;

template <int K, int N>
__global__ void __launch_bounds__(kC1Threads) __attribute__((amdgpu_waves_per_eu(4, 4))) k_conv1x1_stream(Conv1Args a)
{
    constexpr int NB = N / 16, J = K / 16, LDW = N + 4;   // + 4 floats: the four k-groups of a read hit different banks
    __shared__ __attribute__((aligned(16))) float s_w[K * LDW];
    __shared__ __attribute__((aligned(16))) float s_bias[N];
    const int tid = threadIdx.x, lane = tid & 63;
    a.x += (size_t)blockIdx.y * a.xs; a.w += (size_t)blockIdx.y * a.ws; a.y += (size_t)blockIdx.y * a.ys;
    for (int i = tid; i < K * (N / 4); i += kC1Threads) {
        const int k = i / (N / 4), n4 = i - k * (N / 4);
        *reinterpret_cast<float4 *>(s_w + k * LDW + 4 * n4) = *reinterpret_cast<const float4 *>(a.w + (size_t)k * N + 4 * n4);
    }
    for (int i = tid; i < N; i += kC1Threads) s_bias[i] = a.bias ? a.bias[i] : 0.0f;
    __syncthreads();

    const int p_in = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * (kC1Threads / 64) + (tid >> 6), nwaves = gridDim.x * (kC1Threads / 64);
    // weight fragment of (step s = 4 j + c, block nb): s_w[(16 j + 4 q + c) * LDW + nb * 16 + p_in]
    const float *wq = s_w + (4 * q) * LDW + p_in;
    for (int tile = wave; tile < a.tiles; tile += nwaves) {
        // rows are 32-bit here (the entry point refuses more than 2^31 - 16 of them: 64 channels x 4 bytes x 2^31
        // rows would be 550 GB): the clamp is one v_min_i32 against an SGPR -- the 64-bit select kept a VGPR copy
        // of (P - 1)'s high word alive across the loop, the one register the budget did not have
        const int p0 = tile * 16 + p_in, plast = (int)a.P - 1;
        const int64_t p = p0 < plast ? p0 : plast;          // clamped: loads unconditional
        // (the lane's column offset goes through an empty asm: otherwise `a.res + 4 q` and `a.y + 4 q` are hoisted
        // out of the tile loop as two more 64-bit per-lane values, and with 64 accumulator + 16 activation + 8
        // fragment registers live the 128-register budget was 4 short: two scratch reloads per tile)
        int q4 = 4 * q;
        asm volatile("" : "+v"(q4));
        const float *xr = a.x + (p * K + q4);
        f32x4 xv[J];
#pragma unroll
        for (int j = 0; j < J; ++j) xv[j] = *reinterpret_cast<const f32x4 *>(xr + 16 * j);
        f32x4 acc[NB];
        if (a.res) {
            const float *rr = a.res + (p * N + q4);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rr + 16 * nb));
        } else {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 bz = *reinterpret_cast<const f32x4 *>(s_bias + 16 * nb + 4 * q);
            acc[nb] += bz;
        }
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float xb = xv[j][c];
                const float *wrow = wq + (16 * j + c) * LDW;
                // (fragments in batches of at most 8: with all 16 of a 256-wide step live next to the 64
                // accumulator registers the 128-register budget of 4 waves / SIMD was 4-6 registers short)
                constexpr int WB = NB > 8 ? 8 : NB;
#pragma unroll
                for (int n0 = 0; n0 < NB; n0 += WB) {
                    float wf[WB];
#pragma unroll
                    for (int nb = 0; nb < WB; ++nb) wf[nb] = wrow[16 * (n0 + nb)];
#pragma unroll
                    for (int nb = 0; nb < WB; ++nb)
                        acc[n0 + nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nb], xb, acc[n0 + nb], 0, 0, 0);
                    if (n0 + WB < NB) __builtin_amdgcn_sched_barrier(0);
                }
                // the fragment reads of a later step must not climb above this point: fully
                // unrolled, the scheduler hoisted all K * N / 64 of them and spilled
                __builtin_amdgcn_sched_barrier(0);
            }
        if (p0 <= plast) {
            float *yr = a.y + (p * N + q4);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x4 v = acc[nb];
                if (a.relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
                *reinterpret_cast<f32x4 *>(yr + 16 * nb) = v;
            }
        }
    }
}

// The same streaming product for a WIDE output: y[:, n0 : n0 + NT] = relu?(x . W[:, n0 : n0 + NT] + bias
// + residual) with the output row wider than one workgroup's column block (ResNet stage 2: conv3 of a
// bottleneck, 128 -> 512 with the residual, 134 400 pixels at batch 8: 619 MB for 17.6 GFLOP; the
// library GEMM 203 us = 3.0 TB/s).  Grid rows = column blocks of NT = 256 channels; a block's K x NT
// weights (128 x 256: 133 KB with the bank padding) sit in LDS: one workgroup of 16 wavefronts per CU.
// x is read once per column block (69 MB x 2: L2 / MALL-resident behind the first block).
struct WideArgs {
    const float *x, *w, *bias, *res;      // x (P, K), w (K, ldn), bias (ldn) or NULL, res (P, ldn) or NULL
    float *y;                             // (P, ldn)
    int64_t P;
    int32_t relu, tiles, ldn;
};

constexpr int kWideThreads = 1024;

template <int K, int NT>
__global__ void __launch_bounds__(kWideThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) k_conv1x1_wide(WideArgs a)
{
    constexpr int NB = NT / 16, J = K / 16, LDW = NT + 4;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float *s_w = s_dyn, *s_bias = s_w + K * LDW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n0 = blockIdx.y * NT;
    for (int i = tid; i < K * (NT / 4); i += kWideThreads) {
        const int k = i / (NT / 4), n4 = i - k * (NT / 4);
        *reinterpret_cast<float4 *>(s_w + k * LDW + 4 * n4) = *reinterpret_cast<const float4 *>(a.w + (size_t)k * a.ldn + n0 + 4 * n4);
    }
    for (int i = tid; i < NT; i += kWideThreads) s_bias[i] = a.bias ? a.bias[n0 + i] : 0.0f;
    __syncthreads();

    const int p_in = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * (kWideThreads / 64) + (tid >> 6), nwaves = gridDim.x * (kWideThreads / 64);
    const float *wq = s_w + (4 * q) * LDW + p_in;
    for (int tile = wave; tile < a.tiles; tile += nwaves) {
        const int64_t p0 = (int64_t)tile * 16 + p_in;
        const int64_t p = p0 < a.P ? p0 : a.P - 1;         // clamped: loads unconditional
        const float *xr = a.x + p * K + 4 * q;
        f32x4 xv[J];
#pragma unroll
        for (int j = 0; j < J; ++j) xv[j] = *reinterpret_cast<const f32x4 *>(xr + 16 * j);
        f32x4 acc[NB];
        if (a.res) {
            const float *rr = a.res + p * a.ldn + n0 + 4 * q;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rr + 16 * nb));
        } else {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 bz = *reinterpret_cast<const f32x4 *>(s_bias + 16 * nb + 4 * q);
            acc[nb] += bz;
        }
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float xb = xv[j][c];
                const float *wrow = wq + (16 * j + c) * LDW;
                // (fragments in batches of at most 8: with all 16 of a 256-wide step live next to the 64
                // accumulator registers the 128-register budget of 4 waves / SIMD was 4-6 registers short)
                constexpr int WB = NB > 8 ? 8 : NB;
#pragma unroll
                for (int n0 = 0; n0 < NB; n0 += WB) {
                    float wf[WB];
#pragma unroll
                    for (int nb = 0; nb < WB; ++nb) wf[nb] = wrow[16 * (n0 + nb)];
#pragma unroll
                    for (int nb = 0; nb < WB; ++nb)
                        acc[n0 + nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nb], xb, acc[n0 + nb], 0, 0, 0);
                    if (n0 + WB < NB) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        if (p0 < a.P) {
            float *yr = a.y + p * a.ldn + n0 + 4 * q;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x4 v = acc[nb];
                if (a.relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
                *reinterpret_cast<f32x4 *>(yr + 16 * nb) = v;
            }
        }
    }
}

// Two products per pixel tile (ResNet stage 1, the boundary between two bottlenecks, reference
// resnet.py:215-255): y = relu(x . W + bias + residual) (64 -> 256, the block's conv3 + bn3 + add +
// ReLU) is stored AND fed, still in the accumulators, into the next block's conv1 + bn1 + ReLU
// h = relu(y . W2 + bias2) (256 -> 64).  y is written once and not read back: the separate 256 -> 64
// kernel's 550 MB read (batch 8, 200 x 336) disappears.  The accumulator layout of the first product
// (lane (pixel p, group q) holds y[p][16 nb + 4 q + c]) IS the operand layout of the streaming
// kernel above (step 4 j + c consumes k = 16 j + 4 q + c), so the second product needs no shuffle.
// Both weight matrices sit in LDS (66.6 + 69.6 KB): one workgroup of 16 wavefronts per CU.
struct ChainArgs {
    const float *x, *w, *bias, *res;      // x (P, K), w (K, N), bias (N) or NULL, res (P, N) or NULL
    const float *w2, *bias2;              // w2 (N, N2), bias2 (N2) or NULL
    float *y, *h;                         // (P, N), (P, N2)
    int64_t P;
    int32_t tiles;
};

constexpr int kChainThreads = 1024;

template <int K, int N, int N2>
__global__ void __launch_bounds__(kChainThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) k_conv1x1_chain(ChainArgs a)
{
    constexpr int NB = N / 16, J = K / 16, LDW = N + 4, NB2 = N2 / 16, LDW2 = N2 + 4;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float *s_w = s_dyn, *s_w2 = s_w + K * LDW, *s_bias = s_w2 + N * LDW2, *s_bias2 = s_bias + N;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < K * (N / 4); i += kChainThreads) {
        const int k = i / (N / 4), n4 = i - k * (N / 4);
        *reinterpret_cast<float4 *>(s_w + k * LDW + 4 * n4) = *reinterpret_cast<const float4 *>(a.w + (size_t)k * N + 4 * n4);
    }
    for (int i = tid; i < N * (N2 / 4); i += kChainThreads) {
        const int k = i / (N2 / 4), n4 = i - k * (N2 / 4);
        *reinterpret_cast<float4 *>(s_w2 + k * LDW2 + 4 * n4) = *reinterpret_cast<const float4 *>(a.w2 + (size_t)k * N2 + 4 * n4);
    }
    for (int i = tid; i < N; i += kChainThreads) s_bias[i] = a.bias ? a.bias[i] : 0.0f;
    for (int i = tid; i < N2; i += kChainThreads) s_bias2[i] = a.bias2 ? a.bias2[i] : 0.0f;
    __syncthreads();

    const int p_in = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * (kChainThreads / 64) + (tid >> 6), nwaves = gridDim.x * (kChainThreads / 64);
    const float *wq = s_w + (4 * q) * LDW + p_in;
    const float *wq2 = s_w2 + (4 * q) * LDW2 + p_in;
    for (int tile = wave; tile < a.tiles; tile += nwaves) {
        const int64_t p0 = (int64_t)tile * 16 + p_in;
        const int64_t p = p0 < a.P ? p0 : a.P - 1;
        const float *xr = a.x + p * K + 4 * q;
        f32x4 xv[J];
#pragma unroll
        for (int j = 0; j < J; ++j) xv[j] = *reinterpret_cast<const f32x4 *>(xr + 16 * j);
        f32x4 acc[NB];
        if (a.res) {
            const float *rr = a.res + p * N + 4 * q;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rr + 16 * nb));
        } else {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 bz = *reinterpret_cast<const f32x4 *>(s_bias + 16 * nb + 4 * q);
            acc[nb] += bz;
        }
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float xb = xv[j][c];
                const float *wrow = wq + (16 * j + c) * LDW;
                // (fragments in batches of at most 8: with all 16 of a 256-wide step live next to the 64
                // accumulator registers the 128-register budget of 4 waves / SIMD was 4-6 registers short)
                constexpr int WB = NB > 8 ? 8 : NB;
#pragma unroll
                for (int n0 = 0; n0 < NB; n0 += WB) {
                    float wf[WB];
#pragma unroll
                    for (int nb = 0; nb < WB; ++nb) wf[nb] = wrow[16 * (n0 + nb)];
#pragma unroll
                    for (int nb = 0; nb < WB; ++nb)
                        acc[n0 + nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nb], xb, acc[n0 + nb], 0, 0, 0);
                    if (n0 + WB < NB) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        // y = relu(acc): stored, and the operand of the second product
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 v = acc[nb];
            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            acc[nb] = v;
        }
        if (p0 < a.P) {
            float *yr = a.y + p * N + 4 * q;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) *reinterpret_cast<f32x4 *>(yr + 16 * nb) = acc[nb];
        }
        f32x4 acc2[NB2];
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) acc2[nb] = *reinterpret_cast<const f32x4 *>(s_bias2 + 16 * nb + 4 * q);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float xb = acc[j][c];
                const float *wrow = wq2 + (16 * j + c) * LDW2;
                float wf[NB2];
#pragma unroll
                for (int nb = 0; nb < NB2; ++nb) wf[nb] = wrow[16 * nb];
#pragma unroll
                for (int nb = 0; nb < NB2; ++nb)
                    acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nb], xb, acc2[nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        if (p0 < a.P) {
            float *hr = a.h + p * N2 + 4 * q;
#pragma unroll
            for (int nb = 0; nb < NB2; ++nb) {
                f32x4 v = acc2[nb];
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                *reinterpret_cast<f32x4 *>(hr + 16 * nb) = v;
            }
        }
    }
}

}  // namespace ia

extern "C" int ia_conv1x1_chain(const float *x, const float *w, const float *bias, const float *residual,
                                const float *w2, const float *bias2, float *y, float *h, int64_t rows,
                                int k, int n, int n2, void *stream)
{
    if (!x || !w || !w2 || !y || !h || rows < 1) return IA_E_ARG;
    if (((uintptr_t)x & 15u) || ((uintptr_t)w & 15u) || ((uintptr_t)w2 & 15u) || ((uintptr_t)y & 15u) ||
        ((uintptr_t)h & 15u) || ((uintptr_t)residual & 15u))
        return IA_E_ARG;
    if (k != 64 || n != 256 || n2 != 64) return IA_E_ARG;
    ia::ChainArgs a;
    a.x = x; a.w = w; a.bias = bias; a.res = residual; a.w2 = w2; a.bias2 = bias2; a.y = y; a.h = h; a.P = rows;
    const int64_t tiles = (rows + 15) / 16;
    if (tiles > 2147483647LL) return IA_E_ARG;
    a.tiles = (int32_t)tiles;
    int64_t wgs = (tiles + 15) / 16;
    if (wgs > 256) wgs = 256;                              // one resident workgroup per CU (136 KB of LDS)
    const size_t lds = sizeof(float) * (64 * (256 + 4) + 256 * (64 + 4) + 256 + 64);
    hipStream_t s = (hipStream_t)stream;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return IA_E_ARG;
    {   // the attribute belongs to the function on a device: once per device, thread-safe
        static std::mutex mu;
        static std::vector<char> done;
        std::lock_guard<std::mutex> lock(mu);
        if ((size_t)dev >= done.size()) done.resize((size_t)dev + 1, 0);
        if (!done[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ia::k_conv1x1_chain<64, 256, 64>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return ia::hip_status(e);
            done[dev] = 1;
        }
    }
    hipLaunchKernelGGL((ia::k_conv1x1_chain<64, 256, 64>), dim3((unsigned)wgs), dim3(ia::kChainThreads), lds, s, a);
    return ia::hip_status(hipGetLastError());
}

extern "C" int ia_conv1x1_stream(const float *x, const float *w, const float *bias, const float *residual,
                                 float *y, int64_t rows, int k, int n, int relu, void *stream)
{
    if (!x || !w || !y || rows < 1) return IA_E_ARG;
    if (((uintptr_t)x & 15u) || ((uintptr_t)w & 15u) || ((uintptr_t)y & 15u) || ((uintptr_t)residual & 15u)) return IA_E_ARG;
    ia::Conv1Args a;
    a.x = x; a.w = w; a.bias = bias; a.res = residual; a.y = y; a.P = rows; a.relu = relu ? 1 : 0;
    a.xs = a.ws = a.ys = 0;
    const int64_t tiles = (rows + 15) / 16;
    if (rows > 2147483647LL - 16) return IA_E_ARG;          // k_conv1x1_stream indexes rows with 32 bits
    a.tiles = (int32_t)tiles;
    int64_t wgs = (tiles + 7) / 8;
    if (wgs > 512) wgs = 512;                              // two resident workgroups per CU, tiles strided over the wavefronts
    const dim3 grid((unsigned)wgs), block(ia::kC1Threads);
    hipStream_t s = (hipStream_t)stream;
    if (k == 64 && n == 256) hipLaunchKernelGGL((ia::k_conv1x1_stream<64, 256>), grid, block, 0, s, a);
    else if (k == 256 && n == 64) hipLaunchKernelGGL((ia::k_conv1x1_stream<256, 64>), grid, block, 0, s, a);
    else if (k == 64 && n == 64) hipLaunchKernelGGL((ia::k_conv1x1_stream<64, 64>), grid, block, 0, s, a);
    else return IA_E_ARG;
    return ia::hip_status(hipGetLastError());
}

/* The HBM-bound batched products between the Winograd transforms (K, N <= 256 with 16 K weights or
 * fewer per matrix: the 64- and 128-channel bottleneck convolutions, the 48-column reg | iou
 * output): D[b] (rows, n) = A[b] (rows, k) . W[b] (k, n) on the streaming kernel above, one
 * grid row per matrix, its weights in LDS.  The library's GEMMs reach 3.4 TB/s on these shapes.  */
extern "C" int ia_batched_gemm_stream(const float *A, const float *W, float *D, int batch, int64_t rows,
                                      int k, int n, void *stream)
{
    if (!A || !W || !D || rows < 1 || batch < 1 || batch > 65535) return IA_E_ARG;
    if (((uintptr_t)A & 15u) || ((uintptr_t)W & 15u) || ((uintptr_t)D & 15u)) return IA_E_ARG;
    ia::Conv1Args a;
    a.x = A; a.w = W; a.bias = nullptr; a.res = nullptr; a.y = D; a.P = rows; a.relu = 0;
    a.xs = rows * k; a.ws = (int64_t)k * n; a.ys = rows * n;
    const int64_t tiles = (rows + 15) / 16;
    if (rows > 2147483647LL - 16) return IA_E_ARG;          // k_conv1x1_stream indexes rows with 32 bits
    a.tiles = (int32_t)tiles;
    int64_t wgs = (tiles + 7) / 8;                         // every matrix gets its share of the 512 resident workgroups
    const int64_t share = (512 + batch - 1) / batch;
    if (wgs > share) wgs = share;
    if (wgs < 1) wgs = 1;
    const dim3 grid((unsigned)wgs, (unsigned)batch), block(ia::kC1Threads);
    hipStream_t s = (hipStream_t)stream;
    if (k == 64 && n == 64) hipLaunchKernelGGL((ia::k_conv1x1_stream<64, 64>), grid, block, 0, s, a);
    else if (k == 128 && n == 128) hipLaunchKernelGGL((ia::k_conv1x1_stream<128, 128>), grid, block, 0, s, a);
    else if (k == 256 && n == 48) hipLaunchKernelGGL((ia::k_conv1x1_stream<256, 48>), grid, block, 0, s, a);
    else return IA_E_ARG;
    return ia::hip_status(hipGetLastError());
}

/* y = relu?(x . w + bias (+ residual)) for (k, n) = (128, 512) on k_conv1x1_wide (column blocks of 256) */
extern "C" int ia_conv1x1_wide(const float *x, const float *w, const float *bias, const float *residual,
                               float *y, int64_t rows, int k, int n, int relu, void *stream)
{
    if (!x || !w || !y || rows < 1) return IA_E_ARG;
    if (((uintptr_t)x & 15u) || ((uintptr_t)w & 15u) || ((uintptr_t)y & 15u) || ((uintptr_t)residual & 15u)) return IA_E_ARG;
    if (k != 128 || n != 512) return IA_E_ARG;
    ia::WideArgs a;
    a.x = x; a.w = w; a.bias = bias; a.res = residual; a.y = y; a.P = rows; a.relu = relu ? 1 : 0; a.ldn = n;
    const int64_t tiles = (rows + 15) / 16;
    if (tiles > 2147483647LL) return IA_E_ARG;
    a.tiles = (int32_t)tiles;
    int64_t wgs = (tiles + 15) / 16;
    if (wgs > 128) wgs = 128;                              // x 2 column blocks = one resident workgroup per CU
    const size_t lds = sizeof(float) * (128 * (256 + 4) + 256);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return IA_E_ARG;
    {
        static std::mutex mu;
        static std::vector<char> done;
        std::lock_guard<std::mutex> lock(mu);
        if ((size_t)dev >= done.size()) done.resize((size_t)dev + 1, 0);
        if (!done[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ia::k_conv1x1_wide<128, 256>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return ia::hip_status(e);
            done[dev] = 1;
        }
    }
    hipLaunchKernelGGL((ia::k_conv1x1_wide<128, 256>), dim3((unsigned)wgs, 2), dim3(ia::kWideThreads), lds,
                       (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}
