// Geometry of the Winograd F(4x4,3x3) tile list (all pyramid levels of a batch form ONE list of
// 4x4-output tiles: level-major, then image, then row-major) -- shared by the transform kernels
// (wino.hip) and the one-launch Winograd convolution (wino_fused.hip).
#pragma once
#include "ia_internal.hpp"

namespace ia {

struct WinoLevels {
    int32_t L, B;
    int32_t H[IA_MAX_LEVELS], W[IA_MAX_LEVELS], tx[IA_MAX_LEVELS], tpi[IA_MAX_LEVELS];
    int32_t tile_off[IA_MAX_LEVELS + 1];      // prefix of B * tiles_per_image
};

static inline int make_wino_levels(const ia_wino_geom *g, WinoLevels &w)
{
    if (!g || g->num_levels < 1 || g->num_levels > IA_MAX_LEVELS || g->batch < 1) return IA_E_ARG;
    w.L = g->num_levels; w.B = g->batch;
    w.tile_off[0] = 0;
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        const bool on = l < g->num_levels;
        if (on && (g->H[l] < 1 || g->W[l] < 1)) return IA_E_ARG;
        w.H[l] = on ? g->H[l] : 0; w.W[l] = on ? g->W[l] : 0;
        w.tx[l] = (w.W[l] + 3) / 4;
        w.tpi[l] = ((w.H[l] + 3) / 4) * w.tx[l];
        const int64_t n = (int64_t)w.tile_off[l] + (int64_t)g->batch * w.tpi[l];
        if (n > 2147483647LL) return IA_E_ARG;
        w.tile_off[l + 1] = (int32_t)n;
    }
    return 0;
}

struct TileRef { int l, b, y0, x0; };

// Workgroup id -> tile.  Hardware hands consecutive workgroup ids to the 8 XCDs round-robin and
// every XCD has its own L2: with the identity mapping the tiles that share input pixels (a tile
// overlaps its neighbours by two rows / columns, 2.25 reads per pixel) would sit in eight
// different L2s and the overlap would be fetched from HBM again.  Give each XCD one contiguous
// range of tiles instead.
__device__ __forceinline__ int xcd_tile(int bid, int T)
{
    const int per = (T + 7) / 8;
    return (bid & 7) * per + (bid >> 3);
}

// The tables are read with constant indices (scalar loads) and the per-lane level is resolved
// with compares and selects: a per-lane index into a kernel-argument array compiles to vector
// loads from the kernarg segment, and `while (t >= tile_off[l + 1]) ++l` to a chain of them --
// several dependent memory round trips in front of the first pixel load.
__device__ __forceinline__ TileRef locate_tile(const WinoLevels &w, int t, int *Hout = nullptr,
                                               int *Wout = nullptr)
{
    TileRef r;
    int l = 0;
#pragma unroll
    for (int i = 1; i < IA_MAX_LEVELS; ++i) l += (i < w.L && t >= w.tile_off[i]) ? 1 : 0;
    int off = w.tile_off[0], tpi = w.tpi[0], tx = w.tx[0], H = w.H[0], W = w.W[0];
#pragma unroll
    for (int i = 1; i < IA_MAX_LEVELS; ++i) {
        const bool m = l == i;
        off = m ? w.tile_off[i] : off; tpi = m ? w.tpi[i] : tpi; tx = m ? w.tx[i] : tx;
        H = m ? w.H[i] : H; W = m ? w.W[i] : W;
    }
    if (Hout) *Hout = H;
    if (Wout) *Wout = W;
    const int q = t - off;
    r.l = l; r.b = q / tpi;
    const int i = q - r.b * tpi;
    // row-major tiles.  (Strips of 8 tile rows, column-major inside -- meant to shorten the L2
    // reuse distance of the overlapping 6x6 patches -- measured SLOWER: 260 vs 245 us for the
    // head's input transform, 222 vs 209 us for the output transform: neighbouring wavefronts
    // then touch DRAM pages a whole pixel row apart.)
    const int ty = i / tx;
    r.y0 = 4 * ty; r.x0 = 4 * (i - ty * tx);
    return r;
}

// `opaque`: the value stays in scalar registers and the compiler cannot see through it.  Without
// it a chain of selects over table entries is folded back into ONE load at a selected address --
// a vector load from the kernarg segment whose round trip sits in front of the first pixel load.
template <typename P>
__device__ __forceinline__ P opaque(P v)
{
    asm volatile("" : "+s"(v));
    return v;
}
// UNIFORM: l is the same for the whole wavefront (one tile per wavefront) -> a scalar load
template <bool UNIFORM, typename P>
__device__ __forceinline__ P level_ptr(P const (&tab)[IA_MAX_LEVELS], int l)
{
    if (UNIFORM) return tab[l];
    // selected as a byte offset from tab[0]: the asm would hide that a pointer passed through
    // it is a kernel-argument (global) pointer and every access would become flat_load / _store
    const char *base = reinterpret_cast<const char *>(tab[0]);
    int64_t d = 0;
#pragma unroll
    for (int i = 1; i < IA_MAX_LEVELS; ++i) {
        const int64_t di = opaque((int64_t)(reinterpret_cast<const char *>(tab[i]) - base));
        d = (l == i) ? di : d;
    }
    return reinterpret_cast<P>(const_cast<char *>(base + d));
}


constexpr int kMaxSeg = 4;

struct WinoSeg {                          // output channels [c0, c0 + n) -> dst tensors with Cdst channels
    int32_t c0, n, Cdst, coff;
    float *dst[IA_MAX_LEVELS];
};

}  // namespace ia
