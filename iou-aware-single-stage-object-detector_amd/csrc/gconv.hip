// Grouped 3x3 convolution of the ResNeXt bottlenecks (BASELINE config 4: X-101-64x4d; reference
// mmdet/models/backbones/resnext.py:12-91 -- conv2 with groups = 64 / 32 and 4, 8, 16, 32
// channels per group), channels-last fp32, pad 1, stride 1 or 2, folded BatchNorm shift + ReLU
// in the epilogue.
//
// Per group the contraction is tiny (K = 9 * Cg, N = Cg): MIOpen / CK run these layers at
// 230 - 1300 us each (18.8 of the 68 ms of an X-101-64x4d step) although they move 0.14 - 1.1 GB
// -- 25 - 180 us of HBM time.  Here a wavefront owns a "supergroup" of 16 (Cg <= 16) or 32
// (Cg = 32) consecutive channels, i.e. 4 / 2 / 1 whole groups, and walks a row of output pixels
// in tiles of 16:
//   * weights of the supergroup as a dense 16 x 16 (32 x 32) matrix per tap, zero outside the
//     groups' diagonal blocks, BatchNorm scale folded in, pre-arranged per lane and kept in
//     registers for the wavefront's lifetime (36 VGPRs, 144 for Cg = 32);
//   * per tap one 16-byte load per lane -- lane (pixel i, quad kk) reads channels 4kk..4kk+3 of
//     pixel i: 16 pixels x 64 contiguous bytes -- feeds four v_mfma_f32_16x16x4_f32 (the sum
//     over input channels is order free, so component c of the quad is the k-slice of step c);
//   * exact fp32 (the f32 MFMA is a k-ordered fmaf chain), 36 MFMAs per 16 pixels x 16 channels.
// Block-diagonal zero padding wastes MFMA work for Cg = 4 / 8 (4x / 2x), which these layers can
// afford: their cost is the activation traffic.
// MI355X, batch 8 (rocprofv3, X-101-64x4d): layer1 520 us, layer2 275, layer3 153, layer4 140 per
// convolution -- CK: 1300 / 650 / 230 / 230; about half of the f32 MFMA issue rate.  Measured and
// not adopted: an XCD-contiguous row mapping (no change), 2 - 4 independent accumulator chains
// (no change: five wavefronts per SIMD already hide the 40-cycle dependent latency), several rows
// per wavefront to amortise the operand loads (no change).
#include <string.h>
#include "ia_internal.hpp"

namespace ia {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Lanes are (pixel i = lane & 15, channel quad kk = lane >> 4): a DPP row holds the 16 pixels of
// one quad.  value of pixel i-1: row_shr:1 of the centre, lane 0 from the previous tile's lane 15
// (row_ror:1 of it); value of pixel i+1: row_shl:1, lane 15 from the next tile's lane 0
// (row_ror:15 of it).  Lanes without a DPP source keep `old` (bound_ctrl = 0).
__device__ __forceinline__ float dpp_shr1(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        __builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shl1(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        __builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x101, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_ror(float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        0, __builtin_bit_cast(int, src), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ f32x4 shift_from_left(const f32x4 &c, const f32x4 &prev)
{
    f32x4 o;
    o.x = dpp_shr1(dpp_ror<0x121>(prev.x), c.x); o.y = dpp_shr1(dpp_ror<0x121>(prev.y), c.y);
    o.z = dpp_shr1(dpp_ror<0x121>(prev.z), c.z); o.w = dpp_shr1(dpp_ror<0x121>(prev.w), c.w);
    return o;
}
__device__ __forceinline__ f32x4 shift_from_right(const f32x4 &c, const f32x4 &next)
{
    f32x4 o;
    o.x = dpp_shl1(dpp_ror<0x12F>(next.x), c.x); o.y = dpp_shl1(dpp_ror<0x12F>(next.y), c.y);
    o.z = dpp_shl1(dpp_ror<0x12F>(next.z), c.z); o.w = dpp_shl1(dpp_ror<0x12F>(next.w), c.w);
    return o;
}

struct GConvArgs {
    const float *x;            // (B, H, W, C) channels-last
    const float *wpack;        // (SG, 9, NB, 4, NB, 64): per-lane B operands, see pack_grouped_weights
    const float *bias;         // (C) folded BatchNorm shift, or NULL
    float *y;                  // (B, Ho, Wo, C)
    int32_t B, H, W, C, Ho, Wo, SG, relu;
};

// NB = 16-channel blocks per supergroup (1: Cg <= 16, 2: Cg = 32); STRIDE 1 or 2
template <int NB, int STRIDE>
__global__ void __launch_bounds__(256) k_gconv3x3(GConvArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sg = blockIdx.y * 4 + wave;
    if (sg >= a.SG) return;
    const int i = lane & 15, kk = lane >> 4;
    const int row = blockIdx.x;                        // (b, yo)
    const int b = row / a.Ho, yo = row - b * a.Ho;
    const int cbase = sg * 16 * NB;
    // B operands: w[t][ci][c][co] for this lane
    float w[9][NB][4][NB];
    {
        const float *wp = a.wpack + (size_t)sg * 9 * NB * 4 * NB * 64 + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ci = 0; ci < NB; ++ci)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int co = 0; co < NB; ++co)
                        w[t][ci][c][co] = wp[(size_t)(((t * NB + ci) * 4 + c) * NB + co) * 64];
    }
    float bz[NB];
#pragma unroll
    for (int co = 0; co < NB; ++co) bz[co] = a.bias ? a.bias[cbase + 16 * co + i] : 0.0f;
    const float *xb = a.x + (size_t)b * a.H * a.W * a.C + cbase + 4 * kk;
    float *yb = a.y + ((size_t)b * a.Ho + yo) * a.Wo * a.C + cbase + i;
    const int tiles = (a.Wo + 15) / 16;
    // stride 1: the three horizontal taps of a row are the same 16 pixels shifted by one lane.
    // One load per input row and tile instead of three (every load touches 16 half-used cache
    // lines; with nine loads per tile the kernel ran at 85 cycles per MFMA, with three at 75):
    // the dx = -1 / +1 operands come from the centre registers by DPP row shifts, the tile's edge
    // lanes from the previous / next tile's centre (rotated into place), which is prefetched one
    // tile ahead (loads two tiles ahead).
    f32x4 prv[3][NB], cur[3][NB], nxt[3][NB], nn[3][NB];
    auto load_rows = [&](int tile, f32x4 (&dst)[3][NB]) {
        const int xo = tile * 16 + i;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yi = yo + r - 1;
            const bool in = (yi >= 0) && (yi < a.H) && (xo < a.W) && (tile < tiles);
            const int yc = (yi < 0) ? 0 : ((yi >= a.H) ? a.H - 1 : yi);
            const int xc = (xo >= a.W) ? a.W - 1 : xo;
            const float *p = xb + ((size_t)yc * a.W + xc) * a.C;
#pragma unroll
            for (int ci = 0; ci < NB; ++ci) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(p + 16 * ci);   // unconditional
                f32x4 z; z.x = z.y = z.z = z.w = 0.0f;
                dst[r][ci] = in ? q : z;
            }
        }
    };
    if (STRIDE == 1) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int ci = 0; ci < NB; ++ci) prv[r][ci].x = prv[r][ci].y = prv[r][ci].z = prv[r][ci].w = 0.0f;
        load_rows(0, cur);
        if (NB == 1) load_rows(1, nxt);
    }
    for (int tile = 0; tile < tiles; ++tile) {
        f32x4 v[9][NB];
        if (STRIDE == 1) {
            // the right edge lane of THIS tile needs the next tile's centre, so the loads run two
            // tiles ahead: what is requested here is first used in the next iteration
            // (the 32-channel variant has no registers to spare and stays one tile ahead)
            if (NB == 1) load_rows(tile + 2, nn);      // zeros beyond the last tile
            else load_rows(tile + 1, nxt);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int ci = 0; ci < NB; ++ci) {
                    v[3 * r + 1][ci] = cur[r][ci];
                    v[3 * r + 0][ci] = shift_from_left(cur[r][ci], prv[r][ci]);
                    v[3 * r + 2][ci] = shift_from_right(cur[r][ci], nxt[r][ci]);
                }
        } else {
            const int xo = tile * 16 + i;              // this lane's A-operand pixel
#pragma unroll
            for (int t = 0; t < 9; ++t) {              // all loads of the tile first
                const int yi = yo * STRIDE + t / 3 - 1;
                const int xi = xo * STRIDE + t % 3 - 1;
                const bool in = (yi >= 0) && (yi < a.H) && (xi >= 0) && (xi < a.W) && (xo < a.Wo);
                const int yc = (yi < 0) ? 0 : ((yi >= a.H) ? a.H - 1 : yi);
                const int xc = (xi < 0) ? 0 : ((xi >= a.W) ? a.W - 1 : xi);
                const float *p = xb + ((size_t)yc * a.W + xc) * a.C;
#pragma unroll
                for (int ci = 0; ci < NB; ++ci) {
                    const f32x4 q = *reinterpret_cast<const f32x4 *>(p + 16 * ci);   // unconditional
                    f32x4 z; z.x = z.y = z.z = z.w = 0.0f;
                    v[t][ci] = in ? q : z;
                }
            }
        }
        f32x4 acc[NB];
#pragma unroll
        for (int co = 0; co < NB; ++co) { acc[co].x = acc[co].y = acc[co].z = acc[co].w = 0.0f; }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ci = 0; ci < NB; ++ci) {
                const float e[4] = {v[t][ci].x, v[t][ci].y, v[t][ci].z, v[t][ci].w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int co = 0; co < NB; ++co)
                        acc[co] = __builtin_amdgcn_mfma_f32_16x16x4f32(e[c], w[t][ci][c][co], acc[co],
                                                                       0, 0, 0);
            }
        // D: column (lane & 15) = output channel, row 4 * (lane >> 4) + r = pixel of the tile
#pragma unroll
        for (int co = 0; co < NB; ++co) {
            const float o[4] = {acc[co].x, acc[co].y, acc[co].z, acc[co].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int px = tile * 16 + 4 * kk + r;
                float val = o[r] + bz[co];
                if (a.relu) val = (val > 0.0f) ? val : 0.0f;
                if (px < a.Wo) yb[(size_t)px * a.C + 16 * co] = val;
            }
        }
        if (STRIDE == 1) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int ci = 0; ci < NB; ++ci) {
                    prv[r][ci] = cur[r][ci]; cur[r][ci] = nxt[r][ci];
                    if (NB == 1) nxt[r][ci] = nn[r][ci];
                }
        }
    }
}

}  // namespace ia

extern "C" {

// host-side weight arrangement: (C, Cg, 3, 3) grouped weight (+ per-output-channel scale) ->
// (SG, 9, NB, 4, NB, 64) B operands.  Element [sg][t][ci][c][co][lane] = scale[o] * W[o][in][t]
// with o = sg*16*NB + 16*co + (lane & 15), in = 16*ci + 4*(lane >> 4) + c (inside the
// supergroup), zero when o and in belong to different groups.
int ia_grouped_conv3x3_pack(const float *weight, const float *scale, int channels, int groups,
                            float *wpack)
{
    if (!weight || !wpack || channels < 16 || groups < 1 || channels % groups) return IA_E_ARG;
    const int cg = channels / groups;
    if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return IA_E_ARG;
    const int nb = (cg == 32) ? 2 : 1, sgc = 16 * nb;
    if (channels % sgc) return IA_E_ARG;
    const int SG = channels / sgc;
    for (int sg = 0; sg < SG; ++sg)
        for (int t = 0; t < 9; ++t)
            for (int ci = 0; ci < nb; ++ci)
                for (int c = 0; c < 4; ++c)
                    for (int co = 0; co < nb; ++co)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int o = sg * sgc + 16 * co + (lane & 15);
                            const int in = sg * sgc + 16 * ci + 4 * (lane >> 4) + c;
                            float v = 0.0f;
                            if (o / cg == in / cg)
                                v = weight[((size_t)o * cg + (in % cg)) * 9 + t] * (scale ? scale[o] : 1.0f);
                            wpack[((((((size_t)sg * 9 + t) * nb + ci) * 4 + c) * nb + co) * 64) + lane] = v;
                        }
    return 0;
}

int ia_grouped_conv3x3_nhwc(const float *x, const float *wpack, const float *bias, float *y,
                            int batch, int H, int W, int channels, int groups, int stride, int relu,
                            void *stream)
{
    if (!x || !wpack || !y || batch < 1 || H < 1 || W < 1 || channels < 16 || groups < 1 ||
        channels % groups || (stride != 1 && stride != 2))
        return IA_E_ARG;
    if (((uintptr_t)x & 15u) || ((uintptr_t)y & 15u)) return IA_E_ARG;
    const int cg = channels / groups;
    if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return IA_E_ARG;
    const int nb = (cg == 32) ? 2 : 1;
    if (channels % (16 * nb)) return IA_E_ARG;
    ia::GConvArgs a;
    a.x = x; a.wpack = wpack; a.bias = bias; a.y = y;
    a.B = batch; a.H = H; a.W = W; a.C = channels;
    a.Ho = (H + 2 - 3) / stride + 1; a.Wo = (W + 2 - 3) / stride + 1;
    a.SG = channels / (16 * nb); a.relu = relu ? 1 : 0;
    const int64_t rows = (int64_t)batch * a.Ho;
    if (rows > 2147483647LL) return IA_E_ARG;
    dim3 grid((unsigned)rows, (unsigned)((a.SG + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (nb == 1 && stride == 1) hipLaunchKernelGGL((ia::k_gconv3x3<1, 1>), grid, block, 0, s, a);
    else if (nb == 1) hipLaunchKernelGGL((ia::k_gconv3x3<1, 2>), grid, block, 0, s, a);
    else if (stride == 1) hipLaunchKernelGGL((ia::k_gconv3x3<2, 1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((ia::k_gconv3x3<2, 2>), grid, block, 0, s, a);
    return ia::hip_status(hipGetLastError());
}

}  // extern "C"
