// The ResNet stem convolution -- 7x7 / stride 2 / pad 3, 3 -> 64 channels, no bias (reference
// mmdet/models/backbones/resnet.py:403-414 `conv1`, forward :506-512) -- on channels-last fp32 as an
// implicit GEMM on v_mfma_f32_16x16x4_f32.  It was the last convolution of the fp32 network left on
// the library (MIOpen igemm in immediate mode: 474 us + a 69 us zero fill of the 550 MB output its
// split-K kernel accumulates into, batch 8, 800 x 1344).
//
//  * K = (ky, kx, c) = 7 x 21 values per output pixel; with a channels-last 3-channel input the 21
//    values of a kernel row are CONTIGUOUS in memory (and in the LDS patch), so k -> address is
//    ky * row pitch + (k - 21 ky): no im2col.  K is padded 147 -> 148 (37 MFMA steps of 4) with a
//    zero weight row.
//  * A workgroup (4 wavefronts) owns 4 output rows x 64 output columns: the 13 x 133-pixel input
//    patch (20.8 KB) and the whole weight matrix (148 x 64 fp32, rows padded to 80 floats: 47 KB)
//    sit in LDS, two workgroups per CU.  A wavefront computes one output row = four 16-pixel tiles x
//    64 channels (64 accumulator registers): per K step 4 weight reads + 4 patch reads feed 16 MFMAs.
//  * D^T = W^T X^T like csrc/conv1x1_stream.hip: a lane ends up with FOUR CONSECUTIVE OUTPUT CHANNELS
//    of one pixel -- 16-byte stores, 64 contiguous bytes per pixel and channel block.
//  * The output is the RAW convolution: the folded BatchNorm + ReLU + 3x3/2 max-pool stay the one
//    pass of k_affine_relu_maxpool (elementwise.hip).
#include <string.h>
#include "ia_internal.hpp"

namespace ia {

typedef __attribute__((ext_vector_type(4))) float f32x4s;

struct StemArgs {
    const float *x;          // (B, H, W, 3)
    const float *w;          // (148, 64): k = ky * 21 + kx * 3 + c, row 147 zero
    float *y;                // (B, Ho, Wo, 64)
    int32_t B, H, W, Ho, Wo, tiles_x, tiles_y, ntiles;
};

constexpr int kStTR = 4, kStTC = 64;                 // output rows / columns per workgroup
constexpr int kStPR = 2 * kStTR + 5;                 // 13 patch rows
constexpr int kStPCF = (2 * kStTC + 5) * 3;          // 399 floats per patch row
constexpr int kStPS = 404;                           // LDS pitch of a patch row (floats): 3 lead-in + 399 + 2
constexpr int kStLead = 3;                           // the LDS row starts 3 floats early: (2 c0 - 3) * 3 - 3 is a multiple of 4
constexpr int kStK = 148, kStLDW = 80;               // K steps * 4; LDS pitch of a weight row

// AL: image rows 16-byte aligned (W % 4 == 0): the patch is staged with float4 copies -- the LDS row
// starts kStLead floats before the first one a tile needs, at a multiple of four floats of the image
// row; the operand reads (4-byte LDS reads, any alignment) add the lead-in.  Other widths: dword copies.
template <bool AL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_stem_conv7x7s2(StemArgs a)
{
    __shared__ __attribute__((aligned(16))) float s_x[kStPR * kStPS];
    __shared__ __attribute__((aligned(16))) float s_w[kStK * kStLDW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- weights -> LDS ONCE (16-byte pieces: 148 rows x 16); the workgroup then walks its tiles
    // (a workgroup per tile re-read the 47 KB for 9 us of MFMA work: 0.440 -> 0.428 ms at batch 8)
    for (int i = tid; i < kStK * 16; i += 256) {
        const int k = i >> 4, n4 = i & 15;
        *reinterpret_cast<float4 *>(s_w + k * kStLDW + 4 * n4) = *reinterpret_cast<const float4 *>(a.w + k * 64 + 4 * n4);
    }
    const int px = lane & 15, kq = lane >> 4;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int t = tile;
    const int txi = t % a.tiles_x; t /= a.tiles_x;
    const int tyi = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int r0 = tyi * kStTR, c0 = txi * kStTC;
    __syncthreads();                                  // the previous tile's reads of s_x are done
    // ---- input patch -> LDS: rows 2 r0 - 3 .. + 12, floats (2 c0 - 3) * 3 .. + 398 of each row, zero outside
    const float *xb = a.x + (size_t)b * a.H * a.W * 3;
    const int e0 = (2 * c0 - 3) * 3 - kStLead, row_f = a.W * 3;      // first float of the LDS row (a multiple of 4)
    if (AL) {
        constexpr int NV = kStPR * (kStPS / 4);                       // 13 x 101 float4
        constexpr int NIT = (NV + 255) / 256;                         // 6 per thread, all requested first
        float4 pv[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            int i = tid + 256 * u;
            i = i < NV ? i : NV - 1;
            const int pr = i / (kStPS / 4), e = 4 * (i - pr * (kStPS / 4));
            const int yi = 2 * r0 - 3 + pr, xe = e0 + e;              // a multiple of 4, like row_f
            const int yc = yi < 0 ? 0 : (yi >= a.H ? a.H - 1 : yi), xc = xe < 0 ? 0 : (xe >= row_f ? row_f - 4 : xe);
            pv[u] = *reinterpret_cast<const float4 *>(xb + (size_t)yc * row_f + xc);          // clamped, unconditional
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + 256 * u;
            const int pr = i / (kStPS / 4), e = 4 * (i - pr * (kStPS / 4));
            const int yi = 2 * r0 - 3 + pr, xe = e0 + e;
            const bool in = yi >= 0 && yi < a.H && xe >= 0 && xe < row_f;       // whole vectors: xe, row_f multiples of 4
            if (i < NV) reinterpret_cast<float4 *>(s_x)[i] = in ? pv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        constexpr int NIT = (kStPR * kStPS + 255) / 256;              // 21 dwords per thread, all requested first
        float pv[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            int i = tid + 256 * u;
            i = i < kStPR * kStPS ? i : kStPR * kStPS - 1;
            const int pr = i / kStPS, e = i - pr * kStPS;
            const int yi = 2 * r0 - 3 + pr, xe = e0 + e;
            const int yc = yi < 0 ? 0 : (yi >= a.H ? a.H - 1 : yi), xc = xe < 0 ? 0 : (xe >= row_f ? row_f - 1 : xe);
            pv[u] = xb[(size_t)yc * row_f + xc];                      // clamped, unconditional
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + 256 * u;
            const int pr = i / kStPS, e = i - pr * kStPS;
            const int yi = 2 * r0 - 3 + pr, xe = e0 + e;
            const bool in = yi >= 0 && yi < a.H && xe >= 0 && xe < row_f;
            if (i < kStPR * kStPS) s_x[i] = in ? pv[u] : 0.0f;
        }
    }
    __syncthreads();

    // this wavefront: output row r0 + wv; tile mt = columns c0 + 16 mt + px
    const float *xrow = s_x + (2 * wv) * kStPS + (2 * px) * 3 + kStLead;
    const float *wl = s_w + px;
    f32x4s acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mt][nb] = f32x4s{0.0f, 0.0f, 0.0f, 0.0f};
    // operands of step s + 1 are read from LDS before the MFMAs of step s (two register sets; no
    // measurable difference to reading them in front of their step: 0.431 against 0.428 ms)
    float wf[2][4], xv[2][4];
#define ST_READ(S, SET)                                                                             \
    {                                                                                               \
        /* k = 4 s + kq -> (ky, position inside the kernel row); the four lane groups of a step    \
           straddle at most one row boundary */                                                     \
        const int ky0 = (4 * (S)) / 21;                                                             \
        int k = 4 * (S) + kq;                                                                       \
        k = k > 146 ? 146 : k;             /* the padded k = 147: any valid address, its weight is zero */ \
        const int ky = (k >= 21 * (ky0 + 1)) ? ky0 + 1 : ky0;                                       \
        const int off = ky * kStPS + (k - 21 * ky);                                                 \
        _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) wf[SET][nb] = wl[(4 * (S) + kq) * kStLDW + 16 * nb]; \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) xv[SET][mt] = xrow[off + 96 * mt];    /* 16 pixels * 2 * 3 floats */ \
    }
    ST_READ(0, 0)
#pragma unroll
    for (int s = 0; s < kStK / 4; ++s) {
        if (s + 1 < kStK / 4) ST_READ(s + 1, (s + 1) & 1)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s & 1][nb], xv[s & 1][mt], acc[mt][nb], 0, 0, 0);
        // (fully unrolled and left alone, the scheduler would hoist all 296 LDS reads and spill)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef ST_READ
    const int row = r0 + wv;
    if (row < a.Ho) {
        float *yr = a.y + ((size_t)b * a.Ho + row) * a.Wo * 64 + 4 * kq;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int col = c0 + 16 * mt + px;
            if (col < a.Wo) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    *reinterpret_cast<f32x4s *>(yr + (size_t)col * 64 + 16 * nb) = acc[mt][nb];
            }
        }
    }
    }   // tiles of this workgroup
}

// ---- bf16 (BASELINE config 3): the same convolution on v_mfma_f32_16x16x16_bf16.  K = 7 kernel rows x 24
// (one zero, 21 values, two zeros: groups of four consecutive k never straddle a kernel row), 11 steps
// of 16.  The weights live in REGISTERS (11 x 4 fragments x 8 bytes per lane = 88 VGPRs, loaded once
// per wavefront from a fragment-order packing); the 13 x 133-pixel patch sits in LDS as bf16 and a
// lane reads its four consecutive k as two aligned dwords (byte offset 12 px + 2 r, r % 4 == 0).
typedef short v4s __attribute__((ext_vector_type(4)));
constexpr int kSbPitch = 404;                        // bf16 elements per patch row (399 + over-read of the padded k)
constexpr int kSbSteps = 11;

struct StemBfArgs {
    const uint16_t *x;       // (B, H, W, 3) bf16
    const uint16_t *w;       // [11 steps][4 blocks][64 lanes][4] bf16, fragment order
    uint16_t *y;             // (B, Ho, Wo, 64) bf16
    int32_t B, H, W, Ho, Wo, tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ uint32_t stem_bf16_rne(float f)
{
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// AL: the image rows are 4-byte aligned (W even): the patch is staged with aligned DWORD copies.  The
// first element a tile needs, (2 c0 - 3) * 3, is odd, so the LDS row starts ONE ELEMENT EARLIER
// (LDS[i] = row[e0 - 1 + i], e0 - 1 even) and the K grouping is shifted by one to keep the lanes'
// four-element reads dword-aligned in LDS: a kernel row is [zero, 21 values, zero, zero], i.e.
// k' = 1 + kx * 3 + c (see ops.stem_weight_bf16).  Odd W: the same LDS content through 2-byte loads.
template <bool AL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_stem_conv7x7s2_bf16(StemBfArgs a)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_x[kStPR * kSbPitch];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int px = lane & 15, kq = lane >> 4;
    // weight fragments: wfr[s][nb] = W[nb * 16 + px][16 s + 4 kq .. + 3]
    v4s wfr[kSbSteps][4];
#pragma unroll
    for (int s = 0; s < kSbSteps; ++s)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
            wfr[s][nb] = *reinterpret_cast<const v4s *>(a.w + ((size_t)(s * 4 + nb) * 64 + lane) * 4);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int t = tile;
        const int txi = t % a.tiles_x; t /= a.tiles_x;
        const int tyi = t % a.tiles_y;
        const int b = t / a.tiles_y;
        const int r0 = tyi * kStTR, c0 = txi * kStTC;
        __syncthreads();                                  // the previous tile's reads of s_x are done
        const uint16_t *xb = a.x + (size_t)b * a.H * a.W * 3;
        const int e0 = (2 * c0 - 3) * 3 - 1, row_f = a.W * 3;            // first element of the LDS row (even)
        if (AL) {
            constexpr int ND = kStPR * (kSbPitch / 2);                    // 13 x 202 dwords
            constexpr int NIT = (ND + 255) / 256;                         // 11 per thread, all requested first
            uint32_t pv[NIT];
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                int i = tid + 256 * u;
                i = i < ND ? i : ND - 1;
                const int pr = i / (kSbPitch / 2), e = 2 * (i - pr * (kSbPitch / 2));
                const int yi = 2 * r0 - 3 + pr, xe = e0 + e;              // even
                const int yc = yi < 0 ? 0 : (yi >= a.H ? a.H - 1 : yi), xc = xe < 0 ? 0 : (xe >= row_f ? row_f - 2 : xe);
                pv[u] = *reinterpret_cast<const uint32_t *>(xb + (size_t)yc * row_f + xc);      // clamped, unconditional
            }
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int i = tid + 256 * u;
                const int pr = i / (kSbPitch / 2), e = 2 * (i - pr * (kSbPitch / 2));
                const int yi = 2 * r0 - 3 + pr, xe = e0 + e;
                // (row_f and xe are even: a dword is inside or outside the row as a whole)
                const bool in = e < kStPCF + 1 && yi >= 0 && yi < a.H && xe >= 0 && xe < row_f;
                if (i < ND) reinterpret_cast<uint32_t *>(s_x)[i] = in ? pv[u] : 0u;
            }
        } else {
            // 21 values per thread, requested in rounds of 7 (all 21 in flight next to the 88 weight registers were
            // 20 registers over the budget of 2 waves / SIMD: scratch spills in the odd-width variant)
            constexpr int NIT = (kStPR * kSbPitch + 255) / 256, RND = 7;
            static_assert(NIT % RND == 0, "rounds must tile the staging loop");
#pragma unroll 1
            for (int u0 = 0; u0 < NIT; u0 += RND) {
                uint16_t pv[RND];
#pragma unroll
                for (int u = 0; u < RND; ++u) {
                    int i = tid + 256 * (u0 + u);
                    i = i < kStPR * kSbPitch ? i : kStPR * kSbPitch - 1;
                    const int pr = i / kSbPitch, e = i - pr * kSbPitch;
                    const int yi = 2 * r0 - 3 + pr, xe = e0 + e;
                    const int yc = yi < 0 ? 0 : (yi >= a.H ? a.H - 1 : yi), xc = xe < 0 ? 0 : (xe >= row_f ? row_f - 1 : xe);
                    pv[u] = xb[(size_t)yc * row_f + xc];                      // clamped, unconditional
                }
#pragma unroll
                for (int u = 0; u < RND; ++u) {
                    const int i = tid + 256 * (u0 + u);
                    const int pr = i / kSbPitch, e = i - pr * kSbPitch;
                    const int yi = 2 * r0 - 3 + pr, xe = e0 + e;
                    const bool in = e < kStPCF + 1 && yi >= 0 && yi < a.H && xe >= 0 && xe < row_f;
                    if (i < kStPR * kSbPitch) s_x[i] = in ? pv[u] : (uint16_t)0;
                }
            }
        }
        __syncthreads();

        // this wavefront: output row r0 + wv; tile mt = columns c0 + 16 mt + px
        const unsigned char *xrow = reinterpret_cast<const unsigned char *>(s_x) + ((2 * wv) * kSbPitch + 6 * px) * 2;
        f32x4s acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[mt][nb] = f32x4s{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < kSbSteps; ++s) {
            // k = 16 s + 4 kq .. + 3 -> (kernel row, position in the row of 24)
            const int ky0 = (16 * s) / 24;
            int k = 16 * s + 4 * kq;
            k = k > 164 ? 164 : k;                         // the padded k >= 168: any valid address, zero weights
            const int ky = (k >= 24 * (ky0 + 1)) ? ky0 + 1 : ky0;
            const int off = (ky * kSbPitch + (k - 24 * ky)) * 2;            // bytes, a multiple of 8
            v4s xf[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const uint32_t lo = *reinterpret_cast<const uint32_t *>(xrow + off + 192 * mt);
                const uint32_t hi = *reinterpret_cast<const uint32_t *>(xrow + off + 192 * mt + 4);
                xf[mt] = __builtin_bit_cast(v4s, make_uint2(lo, hi));
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wfr[s][nb], xf[mt], acc[mt][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);            // (the LDS reads of later steps stay behind: spills otherwise)
        }
        const int row = r0 + wv;
        if (row < a.Ho) {
            uint16_t *yr = a.y + ((size_t)b * a.Ho + row) * a.Wo * 64 + 4 * kq;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int col = c0 + 16 * mt + px;
                if (col < a.Wo) {
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        const f32x4s v = acc[mt][nb];
                        const uint32_t lo = stem_bf16_rne(v.x) | (stem_bf16_rne(v.y) << 16);
                        const uint32_t hi = stem_bf16_rne(v.z) | (stem_bf16_rne(v.w) << 16);
                        *reinterpret_cast<uint2 *>(yr + (size_t)col * 64 + 16 * nb) = make_uint2(lo, hi);
                    }
                }
            }
        }
    }
}

}  // namespace ia

extern "C" int ia_stem_conv7x7s2_bf16(const void *x, const void *w_packed, void *y, int B, int H, int W,
                                      void *stream)
{
    if (!x || !w_packed || !y || B < 1 || H < 1 || W < 1) return IA_E_ARG;
    if (((uintptr_t)w_packed & 7u) || ((uintptr_t)y & 7u) || ((uintptr_t)x & 1u)) return IA_E_ARG;
    ia::StemBfArgs a;
    a.x = static_cast<const uint16_t *>(x); a.w = static_cast<const uint16_t *>(w_packed); a.y = static_cast<uint16_t *>(y);
    a.B = B; a.H = H; a.W = W;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.tiles_x = (a.Wo + ia::kStTC - 1) / ia::kStTC; a.tiles_y = (a.Ho + ia::kStTR - 1) / ia::kStTR;
    const int64_t tiles = (int64_t)B * a.tiles_x * a.tiles_y;
    if (tiles > 2147483647LL || (int64_t)W * 3 > 2147483647LL) return IA_E_ARG;
    a.ntiles = (int32_t)tiles;
    const int64_t wgs = tiles < 512 ? tiles : 512;
    if ((W & 1) == 0 && ((uintptr_t)x & 3u) == 0)
        hipLaunchKernelGGL(ia::k_stem_conv7x7s2_bf16<true>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(ia::k_stem_conv7x7s2_bf16<false>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}

extern "C" int ia_stem_conv7x7s2(const float *x, const float *w_packed, float *y, int B, int H, int W,
                                 void *stream)
{
    if (!x || !w_packed || !y || B < 1 || H < 1 || W < 1) return IA_E_ARG;
    if (((uintptr_t)w_packed & 15u) || ((uintptr_t)y & 15u) || ((uintptr_t)x & 3u)) return IA_E_ARG;
    ia::StemArgs a;
    a.x = x; a.w = w_packed; a.y = y; a.B = B; a.H = H; a.W = W;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.tiles_x = (a.Wo + ia::kStTC - 1) / ia::kStTC; a.tiles_y = (a.Ho + ia::kStTR - 1) / ia::kStTR;
    const int64_t tiles = (int64_t)B * a.tiles_x * a.tiles_y;
    if (tiles > 2147483647LL || (int64_t)W * 3 > 2147483647LL) return IA_E_ARG;
    a.ntiles = (int32_t)tiles;
    const int64_t wgs = tiles < 512 ? tiles : 512;          // two resident workgroups per CU, tiles strided over them
    if ((W & 3) == 0 && ((uintptr_t)x & 15u) == 0)
        hipLaunchKernelGGL(ia::k_stem_conv7x7s2<true>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(ia::k_stem_conv7x7s2<false>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}
