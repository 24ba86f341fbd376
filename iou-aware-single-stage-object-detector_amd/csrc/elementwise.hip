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
    constexpr int U = 4;                               // packs per thread, all loads issued up front
    using V = typename Pack<T>::V;
    const bool has_res = a.res != nullptr;
    T *x = static_cast<T *>(a.x);
    const T *r = static_cast<const T *>(a.res);
    const uint32_t C = (uint32_t)a.C;
    const int64_t base = ((int64_t)blockIdx.x * U * kEwThreads + threadIdx.x) * N;
    const uint32_t step_mod = (uint32_t)(kEwThreads * N) % C;
    uint32_t c0 = (uint32_t)((uint64_t)base % C);      // one 64-bit modulo per thread
    V xv[U], rv[U];
    int64_t idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        idx[u] = base + (int64_t)u * kEwThreads * N;
        if (idx[u] < total) {
            xv[u] = *reinterpret_cast<const V *>(x + idx[u]);
            if (has_res) rv[u] = *reinterpret_cast<const V *>(r + idx[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (idx[u] < total) {
            float v[N], w[N], sc[N], sh[N], rs[N], rb[N];
            Pack<T>::unpack(xv[u], v);
            if (has_res) Pack<T>::unpack(rv[u], w);
#pragma unroll
            for (int q = 0; q < N; q += 4) {
                const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 s4 = a.scale ? *reinterpret_cast<const float4 *>(a.scale + c0 + q) : one;
                const float4 b4 = a.shift ? *reinterpret_cast<const float4 *>(a.shift + c0 + q) : zero;
                sc[q] = s4.x; sc[q + 1] = s4.y; sc[q + 2] = s4.z; sc[q + 3] = s4.w;
                sh[q] = b4.x; sh[q + 1] = b4.y; sh[q + 2] = b4.z; sh[q + 3] = b4.w;
                if (has_res) {
                    const float4 rs4 = a.rscale ? *reinterpret_cast<const float4 *>(a.rscale + c0 + q) : one;
                    const float4 rb4 = a.rshift ? *reinterpret_cast<const float4 *>(a.rshift + c0 + q) : zero;
                    rs[q] = rs4.x; rs[q + 1] = rs4.y; rs[q + 2] = rs4.z; rs[q + 3] = rs4.w;
                    rb[q] = rb4.x; rb[q + 1] = rb4.y; rb[q + 2] = rb4.z; rb[q + 3] = rb4.w;
                }
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float y = v[j] * sc[j] + sh[j];
                if (has_res) y = y + (w[j] * rs[j] + rb[j]);
                v[j] = (a.relu && !(y > 0.0f)) ? ((y != y) ? y : 0.0f) : y;
            }
            *reinterpret_cast<V *>(x + idx[u]) = Pack<T>::pack(v);
        }
        c0 += step_mod;
        if (c0 >= C) c0 -= C;
    }
}

// Specialised channels-last epilogue (shift always present): which optional operands exist is
// a template parameter, so every load is unconditional, 16 bytes wide and issued before any
// arithmetic.  (The generic kernel above selects per element between "load" and "constant";
// hipcc turns that into a branch and a vmcnt(0) per dword -- 3.7 TB/s instead of 5.)
template <typename T, bool SCALE, bool RES, bool RAFF>
__global__ void __launch_bounds__(kEwThreads) k_affine_act_nhwc_fast(AffineArgs a, int64_t total)
{
    constexpr int N = Pack<T>::N;
    constexpr int U = 4;
    constexpr int Q = N / 4;
    using V = typename Pack<T>::V;
    T *x = static_cast<T *>(a.x);
    const T *r = static_cast<const T *>(a.res);
    const uint32_t C = (uint32_t)a.C;
    const int64_t base = ((int64_t)blockIdx.x * U * kEwThreads + threadIdx.x) * N;
    const uint32_t step_mod = (uint32_t)(kEwThreads * N) % C;
    uint32_t c0 = (uint32_t)((uint64_t)base % C);
    V xv[U], rv[U];
    float4 s4[U][Q], b4[U][Q], rs4[U][Q], rb4[U][Q];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int64_t i = base + (int64_t)u * kEwThreads * N;
        ok[u] = i < total;
        if (!ok[u]) i = total - N;                     // clamped: loads stay unconditional
        xv[u] = *reinterpret_cast<const V *>(x + i);
        if (RES) rv[u] = *reinterpret_cast<const V *>(r + i);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            b4[u][q] = *reinterpret_cast<const float4 *>(a.shift + c0 + 4 * q);
            if (SCALE) s4[u][q] = *reinterpret_cast<const float4 *>(a.scale + c0 + 4 * q);
            if (RES && RAFF) {
                rs4[u][q] = *reinterpret_cast<const float4 *>(a.rscale + c0 + 4 * q);
                rb4[u][q] = *reinterpret_cast<const float4 *>(a.rshift + c0 + 4 * q);
            }
        }
        c0 += step_mod;
        if (c0 >= C) c0 -= C;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float v[N], w[N];
        Pack<T>::unpack(xv[u], v);
        if (RES) Pack<T>::unpack(rv[u], w);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float sh[4] = {b4[u][q].x, b4[u][q].y, b4[u][q].z, b4[u][q].w};
            const float sc[4] = {s4[u][q].x, s4[u][q].y, s4[u][q].z, s4[u][q].w};
            const float rs[4] = {rs4[u][q].x, rs4[u][q].y, rs4[u][q].z, rs4[u][q].w};
            const float rb[4] = {rb4[u][q].x, rb4[u][q].y, rb4[u][q].z, rb4[u][q].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * q + j;
                float y = (SCALE ? v[e] * sc[j] : v[e] * 1.0f) + sh[j];
                if (RES) y = y + (RAFF ? (w[e] * rs[j] + rb[j]) : (w[e] * 1.0f + 0.0f));
                v[e] = (a.relu && !(y > 0.0f)) ? ((y != y) ? y : 0.0f) : y;
            }
        }
        if (ok[u])
            *reinterpret_cast<V *>(x + base + (int64_t)u * kEwThreads * N) = Pack<T>::pack(v);
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
    const int64_t packs = total / n;
    const int64_t blocks = (packs + 4 * ia::kEwThreads - 1) / (4 * ia::kEwThreads);   // 4 packs per thread
    if (blocks > 2147483647LL) return IA_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(ia::kEwThreads);
    const bool raff = residual && res_scale && res_shift;
    const bool fast = shift && total >= n && (!residual || raff || (!res_scale && !res_shift));
#define IA_EW(T, S, R, F) hipLaunchKernelGGL((ia::k_affine_act_nhwc_fast<T, S, R, F>), grid, block, 0, s, a, total)
#define IA_EW_T(T)                                                     \
    do {                                                               \
        if (!fast) hipLaunchKernelGGL(ia::k_affine_act_nhwc<T>, grid, block, 0, s, a, total); \
        else if (scale && residual && raff) IA_EW(T, true, true, true);  \
        else if (scale && residual) IA_EW(T, true, true, false);        \
        else if (scale) IA_EW(T, true, false, false);                   \
        else if (residual && raff) IA_EW(T, false, true, true);         \
        else if (residual) IA_EW(T, false, true, false);                \
        else IA_EW(T, false, false, false);                             \
    } while (0)
    if (dtype == IA_F32) IA_EW_T(float);
    else IA_EW_T(uint16_t);
#undef IA_EW_T
#undef IA_EW
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

// ---------------------------------------------------------------------------
// Stem epilogue: BatchNorm (folded) + ReLU + MaxPool2d(3, stride 2, padding 1) of ResNet
// (reference mmdet/models/backbones/resnet.py:506-512: norm1 -> relu -> maxpool) in one pass over
// the channels-last conv1 output: 550 MB read + 137 MB written per batch-8 step instead of
// (550 r + 550 w) for the epilogue plus (550 r + 137 w) for the pooling.  One lane = four
// channels of one output pixel; relu(max(.)) = max(relu(.)) so the ReLU runs once per output.
namespace ia {

struct PoolArgs {
    const void *x;               // (B, H, W, C) fp32 / bf16
    const float *scale, *shift;  // (C)
    void *out;                   // (B, Ho, Wo, C)
    int32_t B, H, W, C, Ho, Wo;
};

// one thread: 16 bytes of channels (4 fp32 / 8 bf16) of one output pixel; the arithmetic is fp32,
// a bf16 output is rounded once at the end (rounding and ReLU are monotonic: the result equals
// affine -> round -> ReLU -> max-pool on bf16 values, the eager sequence)
template <typename T>
__global__ void __launch_bounds__(256) k_affine_relu_maxpool(PoolArgs a)
{
    constexpr int N = Pack<T>::N;
    using V = typename Pack<T>::V;
    const int cvn = a.C / N;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)a.B * a.Ho * a.Wo * cvn;
    if (gid >= total) return;
    const int c = (int)(gid % cvn) * N;
    int64_t p = gid / cvn;
    const int xo = (int)(p % a.Wo); p /= a.Wo;
    const int yo = (int)(p % a.Ho);
    const int b = (int)(p / a.Ho);
    float s[N], t[N], m[N];
#pragma unroll
    for (int j = 0; j < N; j += 4) {
        const float4 s4 = *reinterpret_cast<const float4 *>(a.scale + c + j);
        const float4 t4 = *reinterpret_cast<const float4 *>(a.shift + c + j);
        s[j] = s4.x; s[j + 1] = s4.y; s[j + 2] = s4.z; s[j + 3] = s4.w;
        t[j] = t4.x; t[j + 1] = t4.y; t[j + 2] = t4.z; t[j + 3] = t4.w;
    }
    const T *x = static_cast<const T *>(a.x);
    V taps[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int y = 2 * yo - 1 + dy;
        const int yc = (y < 0) ? 0 : ((y >= a.H) ? a.H - 1 : y);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = 2 * xo - 1 + dx;
            const int xc = (xx < 0) ? 0 : ((xx >= a.W) ? a.W - 1 : xx);
            // clamped address: the duplicate of an in-range tap never changes a maximum
            taps[dy * 3 + dx] = *reinterpret_cast<const V *>(x + (((size_t)b * a.H + yc) * a.W + xc) * a.C + c);
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) m[j] = -__builtin_inff();
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float v[N];
        Pack<T>::unpack(taps[k], v);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float w = v[j] * s[j] + t[j];
            m[j] = (m[j] < w) ? w : m[j];
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) m[j] = (m[j] > 0.f) ? m[j] : 0.f;
    *reinterpret_cast<V *>(static_cast<T *>(a.out) + (((size_t)b * a.Ho + yo) * a.Wo + xo) * a.C + c) = Pack<T>::pack(m);
}

}  // namespace ia

extern "C" int ia_affine_relu_maxpool_nhwc_dt(const void *x, int dtype, const float *scale, const float *shift,
                                              int B, int H, int W, int C, void *out, void *stream)
{
    if (dtype != IA_F32 && dtype != IA_BF16) return IA_E_ARG;
    const int vec = (dtype == IA_F32) ? 4 : 8;
    if (!x || !scale || !shift || !out || B < 1 || H < 1 || W < 1 || C < vec || (C % vec)) return IA_E_ARG;
    if (((uintptr_t)x & 15u) || ((uintptr_t)out & 15u) || ((uintptr_t)scale & 15u) || ((uintptr_t)shift & 15u))
        return IA_E_ARG;
    ia::PoolArgs a;
    a.x = x; a.scale = scale; a.shift = shift; a.out = out;
    a.B = B; a.H = H; a.W = W; a.C = C;
    a.Ho = (H + 2 - 3) / 2 + 1; a.Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)B * a.Ho * a.Wo * (C / vec);
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 2147483647LL) return IA_E_ARG;
    if (dtype == IA_F32)
        hipLaunchKernelGGL(ia::k_affine_relu_maxpool<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(ia::k_affine_relu_maxpool<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}

extern "C" int ia_affine_relu_maxpool_nhwc(const float *x, const float *scale, const float *shift,
                                           int B, int H, int W, int C, float *out, void *stream)
{
    return ia_affine_relu_maxpool_nhwc_dt(x, IA_F32, scale, shift, B, H, W, C, out, stream);
}

// ---------------------------------------------------------------------------
// FPN top-down step (reference mmdet/models/necks/fpn.py:118-120):
//   laterals[i-1] += F.interpolate(laterals[i], scale_factor=2, mode='nearest')
// in place on channels-last tensors: one read of the coarse map (L2-resident: every coarse pixel
// serves four fine pixels), one read-modify-write of the fine map -- instead of materialising the
// upsampled tensor and adding it in a second pass.
namespace ia {

struct UpAddArgs {
    void *fine;                  // (B, H, W, C) fp32 / bf16
    const void *coarse;          // (B, Hc, Wc, C)
    int32_t B, H, W, Hc, Wc, C;
};

template <typename T>
__global__ void __launch_bounds__(256) k_upsample_add(UpAddArgs a)
{
    constexpr int N = Pack<T>::N;
    using V = typename Pack<T>::V;
    const int cvn = a.C / N;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)a.B * a.H * a.W * cvn;
    if (gid >= total) return;
    const int c = (int)(gid % cvn) * N;
    int64_t p = gid / cvn;
    const int x = (int)(p % a.W); p /= a.W;
    const int y = (int)(p % a.H);
    const int b = (int)(p / a.H);
    // nearest: src = floor(dst * in / out); for the exact factor 2 this is dst >> 1, and in general
    // PyTorch's scale_factor=2 path uses floor(dst * 0.5)
    int ys = y >> 1, xs = x >> 1;
    ys = (ys < a.Hc) ? ys : a.Hc - 1;
    xs = (xs < a.Wc) ? xs : a.Wc - 1;
    V *f = reinterpret_cast<V *>(static_cast<T *>(a.fine) + (((size_t)b * a.H + y) * a.W + x) * a.C + c);
    const V uq = *reinterpret_cast<const V *>(static_cast<const T *>(a.coarse) +
                                             (((size_t)b * a.Hc + ys) * a.Wc + xs) * a.C + c);
    const V vq = *f;
    float u[N], v[N];
    Pack<T>::unpack(uq, u);
    Pack<T>::unpack(vq, v);
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += u[j];             // bf16: one rounding of the fp32 sum, like eager's add
    *f = Pack<T>::pack(v);
}

}  // namespace ia

extern "C" int ia_upsample2x_add_nhwc_dt(void *fine, const void *coarse, int dtype, int B, int H, int W,
                                         int Hc, int Wc, int C, void *stream)
{
    if (dtype != IA_F32 && dtype != IA_BF16) return IA_E_ARG;
    const int vec = (dtype == IA_F32) ? 4 : 8;
    if (!fine || !coarse || B < 1 || H < 1 || W < 1 || Hc < 1 || Wc < 1 || C < vec || (C % vec))
        return IA_E_ARG;
    if (((uintptr_t)fine & 15u) || ((uintptr_t)coarse & 15u)) return IA_E_ARG;
    if (H != 2 * Hc || W != 2 * Wc) return IA_E_ARG;      // scale_factor = 2 exactly (fpn.py:119)
    ia::UpAddArgs a;
    a.fine = fine; a.coarse = coarse; a.B = B; a.H = H; a.W = W; a.Hc = Hc; a.Wc = Wc; a.C = C;
    const int64_t total = (int64_t)B * H * W * (C / vec);
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 2147483647LL) return IA_E_ARG;
    if (dtype == IA_F32)
        hipLaunchKernelGGL(ia::k_upsample_add<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(ia::k_upsample_add<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}

extern "C" int ia_upsample2x_add_nhwc(float *fine, const float *coarse, int B, int H, int W, int Hc,
                                      int Wc, int C, void *stream)
{
    return ia_upsample2x_add_nhwc_dt(fine, coarse, IA_F32, B, H, W, Hc, Wc, C, stream);
}
