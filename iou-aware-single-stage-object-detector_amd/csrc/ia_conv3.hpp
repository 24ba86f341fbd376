// Argument block shared by the bf16 3x3 convolution kernels (conv3x3_bf16.hip: the four-wavefront
// variants; conv3x3_bf16_pp.hip: the 256-pixel ping-pong variant for large maps).
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

// ping-pong variant: 256-pixel tiles, LDS rows (TH + 2) * (TW + 16) of one halo patch buffer
constexpr int kPpMaxRows = 576;
constexpr int kPpTilePx = 256;
int launch_conv3x3_bf16_pp(const Conv3Args &a, dim3 grid, hipStream_t st);

}  // namespace ia
