// Argument block of the bf16 3x3 convolution kernels (conv3x3_bf16.hip; the archived 256-pixel
// ping-pong experiment, tools/experiments/conv3x3_bf16_pingpong_256px.hip.txt, takes the same block).
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

}  // namespace ia
