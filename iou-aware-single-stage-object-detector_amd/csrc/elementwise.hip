// Fused per-channel affine (+ residual) (+ ReLU), in place on an NCHW tensor:
//
//     x[n,c,:] = act( x[n,c,:] * scale[c] + shift[c]  [+ r[n,c,:] * rscale[c] + rshift[c]] )
//
// Inference-time replacement for the eval-mode BatchNorm -> (add) -> ReLU chains
// around the MIOpen convolutions (reference ResNet Bottleneck forward,
// mmdet/models/backbones/resnet.py:215-255; ConvModule conv+bias -> relu,
// mmdet/models/utils/conv_module.py:149-163).  PyTorch eager runs each of those
// as its own full read+write pass over the activation; one pass here does BN3 +
// downsample-BN + residual add + ReLU.  Pure HBM streaming: 16 B per lane,
// one (n,c) plane chunk per workgroup so scale/shift are wave-uniform scalars.
#include "ia_internal.hpp"
#include "ia_math.hpp"

namespace ia {

struct AffineArgs {
    void *x;
    const void *res;
    const float *scale, *shift, *rscale, *rshift;
    int64_t HW;
    int32_t C, relu;
};

constexpr int kEwThreads = 256;
constexpr int kEwPerThread = 16;      // elements per thread per workgroup (4 x 16 B for fp32)

template <typename T> struct Pack;
template <> struct Pack<float> {
    static constexpr int N = 4;
    using V = float4;
    static __device__ __forceinline__ void unpack(const V &q, float (&v)[4]) { v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    static __device__ __forceinline__ V pack(const float (&v)[4]) { return make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Pack<uint16_t> {
    static constexpr int N = 8;
    using V = uint4;
    static __device__ __forceinline__ void unpack(const V &q, float (&v)[8])
    {
        v[0] = from_bits(q.x << 16); v[1] = from_bits(q.x & 0xffff0000u);
        v[2] = from_bits(q.y << 16); v[3] = from_bits(q.y & 0xffff0000u);
        v[4] = from_bits(q.z << 16); v[5] = from_bits(q.z & 0xffff0000u);
        v[6] = from_bits(q.w << 16); v[7] = from_bits(q.w & 0xffff0000u);
    }
    static __device__ __forceinline__ uint32_t rne(float f)       // fp32 -> bf16 bits, round-nearest-even
    {
        uint32_t u = to_bits(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    }
    static __device__ __forceinline__ V pack(const float (&v)[8])
    {
        V q;
        q.x = rne(v[0]) | (rne(v[1]) << 16); q.y = rne(v[2]) | (rne(v[3]) << 16);
        q.z = rne(v[4]) | (rne(v[5]) << 16); q.w = rne(v[6]) | (rne(v[7]) << 16);
        return q;
    }
};

template <typename T> __device__ __forceinline__ void store_f32(T *p, float v);
template <> __device__ __forceinline__ void store_f32<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_f32<uint16_t>(uint16_t *p, float v) { *p = (uint16_t)Pack<uint16_t>::rne(v); }

template <typename T, bool VEC>
__global__ void __launch_bounds__(kEwThreads) k_affine_act(AffineArgs a)
{
    const int plane = blockIdx.x;                 // n * C + c
    const int c = plane % a.C;
    const float sc = a.scale ? a.scale[c] : 1.0f;
    const float sh = a.shift ? a.shift[c] : 0.0f;
    const bool has_res = a.res != nullptr;
    const float rs = (has_res && a.rscale) ? a.rscale[c] : 1.0f;
    const float rb = (has_res && a.rshift) ? a.rshift[c] : 0.0f;
    T *x = static_cast<T *>(a.x) + (size_t)plane * a.HW;
    const T *r = has_res ? static_cast<const T *>(a.res) + (size_t)plane * a.HW : nullptr;
    constexpr int N = Pack<T>::N;
    const int64_t base = (int64_t)blockIdx.y * kEwThreads * kEwPerThread;
    if (VEC) {
        using V = typename Pack<T>::V;
#pragma unroll
        for (int u = 0; u < kEwPerThread / N; ++u) {
            const int64_t i = base + ((int64_t)u * kEwThreads + threadIdx.x) * N;
            if (i < a.HW) {
                float v[N], w[N];
                Pack<T>::unpack(*reinterpret_cast<const V *>(x + i), v);
                if (has_res) Pack<T>::unpack(*reinterpret_cast<const V *>(r + i), w);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float y = v[j] * sc + sh;
                    if (has_res) y = y + (w[j] * rs + rb);
                    v[j] = (a.relu && !(y > 0.0f)) ? ((y != y) ? y : 0.0f) : y;
                }
                *reinterpret_cast<V *>(x + i) = Pack<T>::pack(v);
            }
        }
    } else {
#pragma unroll 4
        for (int u = 0; u < kEwPerThread; ++u) {
            const int64_t i = base + (int64_t)u * kEwThreads + threadIdx.x;
            if (i < a.HW) {
                float y = load_f32<T>(x + i) * sc + sh;
                if (has_res) y = y + (load_f32<T>(r + i) * rs + rb);
                y = (a.relu && !(y > 0.0f)) ? ((y != y) ? y : 0.0f) : y;
                store_f32<T>(x + i, y);
            }
        }
    }
}

// (N, HW, C) -> (N, C, HW) transpose through a padded 64x64 LDS tile: brings channels-last
// head outputs (MIOpen's faster fp32 NHWC convolutions) to the NCHW layout of the head
// kernels.  Reads are contiguous along C, writes contiguous along HW.
template <typename T>
__global__ void __launch_bounds__(256) k_nhwc_to_nchw(const T *src, T *dst, int C, int64_t HW)
{
    __shared__ T tile[64][65];
    const int n = blockIdx.z;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const T *s = src + (size_t)n * HW * C;
    T *d = dst + (size_t)n * HW * C;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
#pragma unroll 4
    for (int j = ty; j < 64; j += 4) {                           // j: position in tile
        const int64_t p = p0 + j;
        const int c = c0 + tx;
        if (p < HW && c < C) tile[j][tx] = s[(size_t)p * C + c];
    }
    __syncthreads();
#pragma unroll 4
    for (int j = ty; j < 64; j += 4) {                           // j: channel in tile
        const int c = c0 + j;
        const int64_t p = p0 + tx;
        if (p < HW && c < C) d[(size_t)c * HW + p] = tile[tx][j];
    }
}

// channels-last variant: x is (N, H, W, C) in memory, the channel is the fastest index
template <typename T>
__global__ void __launch_bounds__(kEwThreads) k_affine_act_nhwc(AffineArgs a, int64_t total)
{
    constexpr int N = Pack<T>::N;
    using V = typename Pack<T>::V;
    const bool has_res = a.res != nullptr;
    T *x = static_cast<T *>(a.x);
    const T *r = static_cast<const T *>(a.res);
    for (int64_t i = ((int64_t)blockIdx.x * kEwThreads + threadIdx.x) * N; i < total;
         i += (int64_t)gridDim.x * kEwThreads * N) {
        const int c0 = (int)(i % a.C);
        float v[N], w[N];
        Pack<T>::unpack(*reinterpret_cast<const V *>(x + i), v);
        if (has_res) Pack<T>::unpack(*reinterpret_cast<const V *>(r + i), w);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int c = c0 + j;
            float y = v[j] * (a.scale ? a.scale[c] : 1.0f) + (a.shift ? a.shift[c] : 0.0f);
            if (has_res)
                y = y + (w[j] * (a.rscale ? a.rscale[c] : 1.0f) + (a.rshift ? a.rshift[c] : 0.0f));
            v[j] = (a.relu && !(y > 0.0f)) ? ((y != y) ? y : 0.0f) : y;
        }
        *reinterpret_cast<V *>(x + i) = Pack<T>::pack(v);
    }
}

}  // namespace ia

extern "C" int ia_nhwc_to_nchw(const void *src, void *dst, int dtype, int N, int C, int64_t HW,
                               void *stream)
{
    if (!src || !dst || N < 1 || C < 1 || HW < 1 || N > 65535) return IA_E_ARG;
    dim3 grid((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)N);
    if (grid.y > 65535) return IA_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == IA_F32)
        hipLaunchKernelGGL(ia::k_nhwc_to_nchw<float>, grid, dim3(256), 0, s,
                           static_cast<const float *>(src), static_cast<float *>(dst), C, HW);
    else if (dtype == IA_BF16)
        hipLaunchKernelGGL(ia::k_nhwc_to_nchw<uint16_t>, grid, dim3(256), 0, s,
                           static_cast<const uint16_t *>(src), static_cast<uint16_t *>(dst), C, HW);
    else return IA_E_ARG;
    return ia::hip_status(hipGetLastError());
}

extern "C" int ia_channel_affine_act_nhwc(void *x, int dtype, const float *scale, const float *shift,
                                          const void *residual, const float *res_scale,
                                          const float *res_shift, int relu, int64_t NHW, int C,
                                          void *stream)
{
    if (!x || NHW < 1 || C < 1) return IA_E_ARG;
    const int n = dtype == IA_F32 ? 4 : (dtype == IA_BF16 ? 8 : 0);
    if (n == 0 || C % n != 0 || ((uintptr_t)x & 15u) || (residual && ((uintptr_t)residual & 15u)))
        return IA_E_ARG;
    ia::AffineArgs a;
    a.x = x; a.res = residual; a.scale = scale; a.shift = shift; a.rscale = res_scale;
    a.rshift = res_shift; a.HW = 0; a.C = C; a.relu = relu;
    const int64_t total = NHW * C;
    int64_t blocks = (total / n + ia::kEwThreads - 1) / ia::kEwThreads;
    blocks = (blocks + 3) / 4;                        // 4 packs per thread
    if (blocks < 1) blocks = 1;
    if (blocks > 65536) blocks = 65536;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == IA_F32)
        hipLaunchKernelGGL(ia::k_affine_act_nhwc<float>, dim3((unsigned)blocks), dim3(ia::kEwThreads), 0, s, a, total);
    else
        hipLaunchKernelGGL(ia::k_affine_act_nhwc<uint16_t>, dim3((unsigned)blocks), dim3(ia::kEwThreads), 0, s, a, total);
    return ia::hip_status(hipGetLastError());
}

extern "C" int ia_channel_affine_act(void *x, int dtype, const float *scale, const float *shift,
                                     const void *residual, const float *res_scale,
                                     const float *res_shift, int relu, int N, int C, int64_t HW,
                                     void *stream)
{
    if (!x || N < 1 || C < 1 || HW < 1 || (int64_t)N * C > 2147483647LL) return IA_E_ARG;
    ia::AffineArgs a;
    a.x = x; a.res = residual; a.scale = scale; a.shift = shift; a.rscale = res_scale;
    a.rshift = res_shift; a.HW = HW; a.C = C; a.relu = relu;
    const int64_t per_block = (int64_t)ia::kEwThreads * ia::kEwPerThread;
    const int64_t chunks = (HW + per_block - 1) / per_block;
    if (chunks > 65535) return IA_E_ARG;
    dim3 grid((unsigned)(N * C), (unsigned)chunks);
    hipStream_t s = (hipStream_t)stream;
    const int esz = dtype == IA_F32 ? 4 : 2;
    const int n = 16 / esz;
    const bool vec = (HW % n == 0) && (((uintptr_t)x & 15u) == 0) &&
                     (!residual || ((uintptr_t)residual & 15u) == 0);
    if (dtype == IA_F32) {
        if (vec) hipLaunchKernelGGL((ia::k_affine_act<float, true>), grid, dim3(ia::kEwThreads), 0, s, a);
        else hipLaunchKernelGGL((ia::k_affine_act<float, false>), grid, dim3(ia::kEwThreads), 0, s, a);
    } else if (dtype == IA_BF16) {
        if (vec) hipLaunchKernelGGL((ia::k_affine_act<uint16_t, true>), grid, dim3(ia::kEwThreads), 0, s, a);
        else hipLaunchKernelGGL((ia::k_affine_act<uint16_t, false>), grid, dim3(ia::kEwThreads), 0, s, a);
    } else return IA_E_ARG;
    return ia::hip_status(hipGetLastError());
}
