// Argument block and epilogue of the bf16 3x3 convolution kernels (conv3x3_bf16.hip; the archived
// 256-pixel ping-pong experiments under tools/experiments/ use the same two).
#pragma once
#include "ia_internal.hpp"

namespace ia {

constexpr int kCvMaxGroups = 2;

// One launch covers a list of feature maps (the pyramid levels of the shared-weight head) and up
// to two groups (the cls / reg towers: different inputs, weights and outputs, one tile list).
struct Conv3Args {
    const uint16_t *x[kCvMaxGroups][IA_MAX_LEVELS];   // (B, H_l, W_l, .) bf16, pixel stride xs
    uint16_t *y[kCvMaxGroups][IA_MAX_LEVELS];         // (B, H_l, W_l, .) bf16, pixel stride ys
    const uint16_t *wp;                   // packed weights [groups][ntile][Cin / 32][9][256][32]
    const float *bias;                    // (groups * Cout) or NULL
    int32_t L, B, Cin, Cout, xs, ys, relu, ntile;     // Cin / Cout per group; ntile = ceil(Cout / 256)
    int32_t H[IA_MAX_LEVELS], W[IA_MAX_LEVELS], TH[IA_MAX_LEVELS], TW[IA_MAX_LEVELS];
    int32_t tiles_y[IA_MAX_LEVELS], tiles_x[IA_MAX_LEVELS], tile_off[IA_MAX_LEVELS + 1];
};

typedef float conv3_f32x16 __attribute__((ext_vector_type(16)));
typedef float conv3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 conv3_bf16x2 __attribute__((ext_vector_type(2)));

// Epilogue of a wavefront: MB x 2 accumulator blocks of 32 pixels x 32 channels (block (i, j): column
// n = lane & 31, row (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) -> + bias, ReLU, ONE rounding to bf16
// (v_cvt_pk_bf16_f32, round to nearest even), stores.  Neighbouring lanes hold neighbouring channels of
// one pixel: lane pairs swap one value (DPP) and store 4 bytes each, 64-byte runs per pixel.
// The rows a lane stores follow each other at +2, +6, +2, +6 ... pixels of the tile, so (row, column,
// element offset) advance by additions and one wrap test per row -- round 4's epilogue divided by the
// tile width for every store (64 integer divisions per lane: 10-20 us of vector work per tile, the
// largest part of a tile's time outside its K loop, profiles/r05_conv3x3_pp_variants.txt).
//   yimg: the image's output (y + b * H * W * ys); m_wave: first tile pixel of this wavefront's rows;
//   n_base: first output channel of its 64 (inside the group); bias_off: group * Cout
template <int MB>
__device__ __forceinline__ void conv3_store_tile(const conv3_f32x16 (&acc)[MB][2], const Conv3Args &a, uint16_t *yimg,
                                                 int m_wave, int n_base, int bias_off, int H, int W, int TW,
                                                 int y0, int x0, int tile_px, int lane)
{
    const int odd = lane & 1;
    const int n0 = n_base + (lane & 31), n1 = n0 + 32;
    const bool ok0 = (n0 - odd) + 1 < a.Cout, ok1 = (n1 - odd) + 1 < a.Cout;     // the channel pair this lane stores (Cout is even)
    const float bz0 = (a.bias && n0 < a.Cout) ? a.bias[bias_off + n0] : 0.0f;
    const float bz1 = (a.bias && n1 < a.Cout) ? a.bias[bias_off + n1] : 0.0f;
    int m = m_wave + 4 * (lane >> 5) + odd;       // even lanes store rows r, odd lanes rows r + 1
    int ty = m / TW, tx = m - ty * TW;            // the one division of the epilogue
    int off = ((y0 + ty) * W + (x0 + tx)) * a.ys + (n0 - odd);
    const int wrap = (W - TW) * a.ys;
    const bool relu = a.relu != 0;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const bool live = m < tile_px && y0 + ty < H && x0 + tx < W;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bz = j ? bz1 : bz0;
                float v0 = acc[i][j][r] + bz, v1 = acc[i][j][r + 1] + bz;
                if (relu) { v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; }
                // even lane keeps row r and takes the odd neighbour's row-r value (channel n + 1);
                // odd lane keeps row r + 1 and takes the even neighbour's (channel n - 1)
                const float give = odd ? v0 : v1;
                const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
                const conv3_f32x2 pr = {odd ? got : v0, odd ? v1 : got};
                const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, conv3_bf16x2));
                if (live && (j ? ok1 : ok0)) *reinterpret_cast<uint32_t *>(yimg + off + j * 32) = pk;
            }
            constexpr int d0 = 2, d1 = 6;
            const int d = (r & 2) ? d1 : d0;      // compile-time per unrolled iteration
            m += d; tx += d; off += d * a.ys;
            if (tx >= TW) { tx -= TW; ++ty; off += wrap; }
            if ((r & 2) && tx >= TW) { tx -= TW; ++ty; off += wrap; }      // +6 over a tile narrower than 6
        }
    }
}

}  // namespace ia
