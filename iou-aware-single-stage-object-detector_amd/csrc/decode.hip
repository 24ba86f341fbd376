// Streaming stages of the inference head path (HBM-bound):
//
//   k_rowmax  -- reads the (B, A*C, H, W) class logits and (B, A, H, W) IoU
//                logits of every level ONCE, coalesced along W, and writes one
//                fused row-max score per anchor
//                (reference iou_aware_retina_head.py:502-513,528-531,539).
//   k_gather  -- for the selected candidates only: regenerate the anchor from
//                its index, delta2bbox, clamp, rescale; emit the 80 fused
//                scores class-major (reference :545-558, transforms.py:44-78,
//                anchor_generator.py:53-70).
//
// Layout notes (gfx950): a wavefront owns one anchor `a` and a run of 256
// (fp32) / 512 (bf16) consecutive positions of one channel plane, so every
// global load instruction is a contiguous 1 KiB segment, 16 B per lane.
// Measured on MI355X (tools/ubench/rowmax_variants.hip): single-wavefront
// workgroups with the tile index fastest stream at 5.4 TB/s, more than
// 9-wavefront workgroups with an LDS transpose of the output (4.7 TB/s).  The
// row-max array is stored anchor-major per level ((A, HW) blocks) so each lane
// writes consecutive floats: stride-A stores showed 6.6x write amplification in
// WRITE_SIZE (42.5 MB vs 6.45 MB per batch-8 launch).
// sqrt(sigmoid(.)) is monotone non-decreasing in
// fp32 (exhaustively checked by tests/test_oracle_math.py), so the max over
// classes is taken on the raw logits and the transcendental part runs once
// per anchor instead of once per class.
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_rowmax_dev.hpp"

namespace ia {

struct RowmaxArgs {
    LevelTable t;
    ia_level_ptrs p;
    float *rowmax;
    int32_t blk_off[IA_MAX_LEVELS + 1];   // prefix of tiles_l * A, levels in REVERSE order
    int32_t blocks_per_img;
    int32_t anchors_per_img;
    // first step of the top-k (select.hip): maxima of groups of consecutive stored scores;
    // groupmax == nullptr: not wanted
    SelPlan plan;
    uint32_t *groupmax;                   // ordered keys (bits | 0x80000000), like the channels-last kernel
};

// One wavefront per (image, level, anchor, tile of 64*PPL positions); tile is the
// fastest-varying block coordinate, so wavefronts that are resident together
// stream neighbouring 1 KiB pieces of the same class planes.  Small levels are
// mapped to the lowest block ids so their (latency-bound) work overlaps the big
// levels' streaming instead of forming a tail.  Loads are never predicated:
// out-of-range lanes read a clamped in-range address and drop the result, which
// keeps the class loop a straight, software-pipelined run of independent loads.
template <typename T>
__global__ void __launch_bounds__(64) k_rowmax(RowmaxArgs a)
{
    constexpr int PPL = Lane<T>::PPL;
    constexpr int TILE = 64 * PPL;
    const int lane = threadIdx.x;
    const int A = a.t.A, C = a.t.C, L = a.t.num_levels;
    const int b = blockIdx.x / a.blocks_per_img;
    int rem = blockIdx.x - b * a.blocks_per_img;
    int rl = 0;
    while (rem >= a.blk_off[rl + 1]) ++rl;
    rem -= a.blk_off[rl];
    const int l = L - 1 - rl;
    const int HW = a.t.H[l] * a.t.W[l];
    const int tiles = (HW + TILE - 1) / TILE;
    const int an = rem / tiles;
    const int p0 = (rem - an * tiles) * TILE;
    const T *cls = static_cast<const T *>(a.p.cls[l]) + ((size_t)b * A + an) * C * HW;
    const T *iou = static_cast<const T *>(a.p.iou[l]) + ((size_t)b * A + an) * HW;
    // row-max array: per level an (A, HW) block, anchor-major -> contiguous, full-line stores
    float *out = a.rowmax + (size_t)b * a.anchors_per_img + a.t.anchor_off[l] + (size_t)an * HW;

    const float ninf = -__builtin_inff();
    float m[PPL], il[PPL];
    int pos[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) m[j] = ninf;
    if (HW % PPL == 0) {
        const int pp = p0 + lane * PPL;
        const int pc = (pp < HW) ? pp : (HW - PPL);        // clamped, still 16 B aligned
#pragma unroll
        for (int j = 0; j < PPL; ++j) pos[j] = pp + j;
        const T *src = cls + pc;
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            float v[PPL];
            Lane<T>::load(src + (size_t)c * HW, v);
#pragma unroll
            for (int j = 0; j < PPL; ++j) m[j] = (m[j] < v[j]) ? v[j] : m[j];
        }
        Lane<T>::load(iou + pc, il);
    } else {
        int pc[PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            pos[j] = p0 + lane + 64 * j;
            pc[j] = (pos[j] < HW) ? pos[j] : (HW - 1);
        }
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const T *src = cls + (size_t)c * HW;
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                float v = load_f32<T>(src + pc[j]);
                m[j] = (m[j] < v) ? v : m[j];
            }
        }
#pragma unroll
        for (int j = 0; j < PPL; ++j) il[j] = load_f32<T>(iou + pc[j]);
    }
    float sc[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        sc[j] = sqrt_sigmoidf_(m[j]) * sqrt_sigmoidf_(il[j]);
        if (pos[j] < HW) out[pos[j]] = sc[j];
        else sc[j] = 0.0f;                                  // scores are >= 0
    }
    if (a.groupmax) {
        const int g = a.plan.grp[l];
        if (g) {
            // groups of g consecutive positions of this (image, anchor) plane
            const int gpp = (HW + g - 1) / g;
            uint32_t *gm = a.groupmax + a.plan.goff[l] + ((size_t)b * A + an) * gpp;
            if (HW % PPL == 0) {                            // lane = PPL consecutive positions
                if (g >= PPL) {
                    float v = sc[0];
#pragma unroll
                    for (int j = 1; j < PPL; ++j) v = (v < sc[j]) ? sc[j] : v;
                    const int lanes = g / PPL;
                    v = lanes_max(v, lanes);
                    if ((lane & (lanes - 1)) == 0 && pos[0] < HW) gm[pos[0] / g] = to_bits(v) | 0x80000000u;
                } else {                                    // g = 4, PPL = 8: two groups per lane
#pragma unroll
                    for (int h = 0; h < PPL; h += 4) {
                        float v = sc[h];
#pragma unroll
                        for (int j = 1; j < 4; ++j) v = (v < sc[h + j]) ? sc[h + j] : v;
                        if (pos[h] < HW) gm[pos[h] / g] = to_bits(v) | 0x80000000u;
                    }
                }
            } else {                                        // lane = positions p0 + lane + 64 j
#pragma unroll
                for (int j = 0; j < PPL; ++j) {
                    const float v = lanes_max(sc[j], g);
                    if ((lane & (g - 1)) == 0 && pos[j] < HW) gm[pos[j] / g] = to_bits(v) | 0x80000000u;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Channels-last head outputs (what MIOpen's NHWC convolutions write): a level is one flat
// array of B*HW*A rows of C contiguous logits -- the reference's own
// permute(0,2,3,1).reshape(-1, C) view (:502-507) without the copy.  One wavefront takes 64
// consecutive rows = one contiguous 64*C*sizeof(T)-byte run (20 KiB for C = 80 fp32) with VPR
// fully coalesced, non-temporal 16-byte loads per lane (1 KiB per instruction); lane maxima
// are transposed through LDS (row stride VPR+1: conflict-free) so that lane r reduces row r,
// and the row maxima are stored in row order p*A + a, 256 B per wavefront.
template <typename T, int VPR_T>         // VPR_T = 0: run-time vectors per row
__global__ void __launch_bounds__(64) k_rowmax_nhwc(RowmaxNhwcArgs a)
{
    __shared__ float s_m[64 * ((VPR_T ? VPR_T : kMaxVpr) + 1)];
    int rem = blockIdx.x, rl = 0;
    while (rem >= a.blk_off[rl + 1]) ++rl;
    rem -= a.blk_off[rl];
    const int l = a.big_first ? rl : a.t.num_levels - 1 - rl;
    rowmax_nhwc_wave<T, VPR_T, false>(a, l, rem, s_m, (int)threadIdx.x);
}

// unused dynamic LDS per workgroup = an occupancy cap for the streaming kernel (tools/ubench/
// rowmax_bench.hip sweeps it)
int rowmax_nhwc_lds_pad = 0;
int rowmax_nhwc_big_first = 1;           // block order: largest level first (-2 us: the latency-bound small levels fill the tail)

static int launch_rowmax_nhwc(const LevelTable &t, const ia_level_ptrs &p, int batch, int dtype,
                              float *rowmax, hipStream_t s, float *groupmax)
{
    const int ppl = (dtype == IA_F32) ? Lane<float>::PPL : Lane<uint16_t>::PPL;
    if (t.C % ppl != 0 || t.C / ppl > kMaxVpr) return IA_E_ARG;
    RowmaxNhwcArgs a;
    a.t = t; a.p = p; a.rowmax = rowmax; a.batch = batch;
    a.anchors_per_img = t.anchor_off[t.num_levels];
    a.groupmax = reinterpret_cast<uint32_t *>(groupmax);
    int prc = make_sel_plan(t, batch, a.plan);
    if (prc) return prc;
    a.blk_off[0] = 0;
    a.big_first = rowmax_nhwc_big_first;
    for (int rl = 0; rl < IA_MAX_LEVELS; ++rl) {
        const int l = a.big_first ? (rl < t.num_levels ? rl : -1) : t.num_levels - 1 - rl;
        int64_t n = 0;
        if (l >= 0) {
            if (((uintptr_t)p.cls[l] & 15u) != 0) return IA_E_ARG;      // 16-byte vector loads
            n = (int64_t)batch * sel_units_per_image(t, l);
        }
        if (a.blk_off[rl] + n > 2147483647LL) return IA_E_ARG;
        a.blk_off[rl + 1] = a.blk_off[rl] + (int32_t)n;
    }
    dim3 grid((unsigned)a.blk_off[t.num_levels]);
    const int vpr = t.C / ppl;
    const size_t pad = (size_t)rowmax_nhwc_lds_pad;
    if (dtype == IA_F32) {
        if (vpr == 20) hipLaunchKernelGGL((k_rowmax_nhwc<float, 20>), grid, dim3(64), pad, s, a);
        else hipLaunchKernelGGL((k_rowmax_nhwc<float, 0>), grid, dim3(64), pad, s, a);
    } else {
        if (vpr == 10) hipLaunchKernelGGL((k_rowmax_nhwc<uint16_t, 10>), grid, dim3(64), pad, s, a);
        else hipLaunchKernelGGL((k_rowmax_nhwc<uint16_t, 0>), grid, dim3(64), pad, s, a);
    }
    return hip_status(hipGetLastError());
}

// ---------------------------------------------------------------------------
// use_sigmoid_cls = False (reference iou_aware_retina_head.py:506-507,540-541): the class tensor
// carries C + 1 channels per anchor, channel 0 = background; scores = softmax over all C + 1, fused
// with the IoU prediction like the sigmoid scores (:531), the row maximum over the FOREGROUND
// columns only.  Canonical arithmetic (the oracle restates the same sequence): m = max over the
// C + 1 logits; s = sum of exp(x_c - m), added in class order; score_c = sqrt(exp(x_c - m) / s) *
// sqrt(sigmoid(iou)).  Correctly rounded division, square root and the product with a value >= 0
// are non-decreasing, so the maximum over c of score_c is the score of the largest exp(x_c - m).
// None of the four IoU-aware configs takes this branch: one thread per anchor row, both memory
// orders, no tuning -- a correct path, not a streaming kernel.
struct SoftmaxRowArgs {
    LevelTable t;
    ia_level_ptrs p;
    float *rowmax;
    int32_t anchors_per_img;
};

template <typename T>
__device__ __forceinline__ void softmax_row_stats(const T *x, size_t cs, int Cin, float &m, float &s, float &e_fg)
{
    m = -__builtin_inff();
    for (int c = 0; c < Cin; ++c) {
        const float v = load_f32<T>(x + (size_t)c * cs);
        m = (m < v) ? v : m;
    }
    s = 0.0f; e_fg = 0.0f;
    for (int c = 0; c < Cin; ++c) {
        const float e = expf_(load_f32<T>(x + (size_t)c * cs) - m);
        s = s + e;
        if (c >= 1) e_fg = (e_fg < e) ? e : e_fg;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_rowscore_softmax(SoftmaxRowArgs a)
{
    const int l = (int)blockIdx.y % a.t.num_levels, b = (int)blockIdx.y / a.t.num_levels;
    const int A = a.t.A, Cin = a.t.C + 1, HW = a.t.H[l] * a.t.W[l];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // position in the level's stored order
    if (i >= (int64_t)HW * A) return;
    const bool nhwc = a.t.layout == IA_LAYOUT_NHWC;
    const int pos = nhwc ? (int)(i / A) : (int)(i % HW);
    const int an = nhwc ? (int)(i % A) : (int)(i / HW);
    const size_t row = ((size_t)b * HW + pos) * A + an;
    const size_t cs = nhwc ? (size_t)1 : (size_t)HW;
    const T *cls = static_cast<const T *>(a.p.cls[l]) + (nhwc ? row * Cin : ((size_t)b * A + an) * Cin * HW + pos);
    const T *iou = static_cast<const T *>(a.p.iou[l]) + (nhwc ? row : ((size_t)b * A + an) * HW + pos);
    float m, s, e;
    softmax_row_stats<T>(cls, cs, Cin, m, s, e);
    a.rowmax[(size_t)b * a.anchors_per_img + a.t.anchor_off[l] + i] =
        __builtin_sqrtf(e / s) * sqrt_sigmoidf_(load_f32<T>(iou));
}

static int launch_rowscore_softmax(const LevelTable &t, const ia_level_ptrs &p, int batch, int dtype,
                                   float *rowmax, hipStream_t s)
{
    SoftmaxRowArgs a;
    a.t = t; a.p = p; a.rowmax = rowmax; a.anchors_per_img = t.anchor_off[t.num_levels];
    int64_t nmax = 1;
    for (int l = 0; l < t.num_levels; ++l) {
        const int64_t n = t.anchor_off[l + 1] - t.anchor_off[l];
        nmax = n > nmax ? n : nmax;
    }
    if ((int64_t)batch * t.num_levels > 65535) return IA_E_ARG;
    const dim3 grid((unsigned)((nmax + 255) / 256), (unsigned)(batch * t.num_levels));
    if (dtype == IA_F32) hipLaunchKernelGGL(k_rowscore_softmax<float>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_rowscore_softmax<uint16_t>, grid, dim3(256), 0, s, a);
    return hip_status(hipGetLastError());
}

int launch_rowmax(const LevelTable &t, const ia_level_ptrs &p, int batch, int dtype, float *rowmax,
                  hipStream_t s, float *groupmax)
{
    if (batch < 1 || !rowmax) return IA_E_ARG;
    if (dtype != IA_F32 && dtype != IA_BF16) return IA_E_ARG;
    if (t.softmax) {
        // (group maxima, when wanted, are derived from the finished array by the caller:
        // launch_groupmax -- this kernel does not emit them)
        (void)groupmax;
        return launch_rowscore_softmax(t, p, batch, dtype, rowmax, s);
    }
    if (t.layout == IA_LAYOUT_NHWC)
        return launch_rowmax_nhwc(t, p, batch, dtype, rowmax, s, groupmax);
    const int tile = 64 * (dtype == IA_F32 ? Lane<float>::PPL : Lane<uint16_t>::PPL);
    RowmaxArgs a;
    a.t = t; a.p = p; a.rowmax = rowmax;
    a.groupmax = reinterpret_cast<uint32_t *>(groupmax);
    int prc = make_sel_plan(t, batch, a.plan);
    if (prc) return prc;
    a.blk_off[0] = 0;
    for (int rl = 0; rl < IA_MAX_LEVELS; ++rl) {
        const int l = t.num_levels - 1 - rl;
        int n = 0;
        if (l >= 0) n = (t.H[l] * t.W[l] + tile - 1) / tile * t.A;
        a.blk_off[rl + 1] = a.blk_off[rl] + n;
    }
    a.blocks_per_img = a.blk_off[t.num_levels];
    a.anchors_per_img = t.anchor_off[t.num_levels];
    dim3 grid((unsigned)(a.blocks_per_img * batch));
    if (dtype == IA_F32) hipLaunchKernelGGL(k_rowmax<float>, grid, dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_rowmax<uint16_t>, grid, dim3(64), 0, s, a);
    return hip_status(hipGetLastError());
}

}  // namespace ia

#include "ia_gather_dev.hpp"

namespace ia {

constexpr int kGroups = 4;   // class groups per candidate (threadIdx.y)

template <typename T>
__global__ void __launch_bounds__(64 * kGroups) k_gather(GatherArgs a)
{
    __shared__ float gmax[kGroups][64];
    const int lane = threadIdx.x, grp = threadIdx.y;
    const int b = blockIdx.y;
    const int r_raw = blockIdx.x * 64 + lane;
    const bool live = r_raw < a.R;
    const int r = live ? r_raw : a.R - 1;        // clamp: every thread reaches the barrier
    const int A = a.t.A, C = a.t.C;
    const LevelSel lv = level_of_candidate(a, r);
    const int l = lv.l;
    const int W = lv.W, HW = lv.H * W;
    const int idx = a.cand_idx[(size_t)b * a.R + r];
    const int pos = idx / A, an = idx - pos * A;
    // channel stride / element offsets of candidate (pos, an) in either memory order
    const bool nhwc = a.t.layout == IA_LAYOUT_NHWC;
    const size_t cs = nhwc ? (size_t)1 : (size_t)HW;
    const size_t row = ((size_t)b * HW + pos) * A + an;                  // (B, HW, A) row id
    const T *cls = static_cast<const T *>(lv.cls) +
                   (nhwc ? row * C : ((size_t)b * A + an) * C * HW + pos);
    const T *iou = static_cast<const T *>(lv.iou) +
                   (nhwc ? row : ((size_t)b * A + an) * HW + pos);
    const float sq_iou = sqrt_sigmoidf_(load_f32<T>(iou));
    const int cpg = (C + kGroups - 1) / kGroups;
    const int c0 = grp * cpg;
    const int c1 = (c0 + cpg < C) ? (c0 + cpg) : C;
    float *so = a.scores_t + (size_t)b * C * a.Rs + r;
    float best = 0.0f;                           // scores are >= 0
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        float x = load_f32<T>(cls + (size_t)c * cs);
        float sc = sqrt_sigmoidf_(x) * sq_iou;
        if (live) so[(size_t)c * a.Rs] = sc;
        best = (best < sc) ? sc : best;
    }
    gmax[grp][lane] = best;
    __syncthreads();
    if (grp == 0 && live) {
        if (a.best_score) {
#pragma unroll
            for (int g2 = 1; g2 < kGroups; ++g2) best = (best < gmax[g2][lane]) ? gmax[g2][lane] : best;
            a.best_score[(size_t)b * a.R + r] = best;
        }
        const T *reg = static_cast<const T *>(lv.reg) +
                       (nhwc ? row * 4 : ((size_t)b * A + an) * 4 * HW + pos);
        const float *ba = a.ba.v[l][an];
        reinterpret_cast<float4 *>(a.boxes)[(size_t)b * a.R + r] =
            decode_box(a, b, ba[0], ba[1], ba[2], ba[3], W, lv.stride, pos, load_f32<T>(reg), load_f32<T>(reg + cs),
                       load_f32<T>(reg + 2 * cs), load_f32<T>(reg + 3 * cs));
    }
}

// ---------------------------------------------------------------------------
// Channels-last head outputs: a candidate's C logits are one contiguous run (320 B for C = 80
// fp32).  A workgroup takes kGTile = 32 candidates with vpr / VT threads per candidate (vpr = C *
// sizeof(T) / 16 vectors per row, VT = 2 of them per thread when vpr is even): every thread loads
// its 16-byte vectors (a wavefront instruction reads whole rows), turns the logits into fused
// scores, and the scores go through an LDS tile so that the class-major score rows are written as
// contiguous 128-byte runs; the first thread of a candidate also decodes its box.  At batch 8:
// 1 176 workgroups of 5 wavefronts, all resident at once (64 SGPRs: the per-level scalars of 5
// levels; with 8 levels' worth, 90 SGPRs, only 4 workgroups fit a CU and the tail ran as a second
// round).  The thread-per-20-classes kernel before it: 19 us.
constexpr int kGTile = 32;

#ifdef IA_GATHER_PROFILE
__device__ unsigned long long g_gather_prof[4][16];
__device__ unsigned long long g_gather_blk[8][160][4];      // per block: start, end of wave 0, end of last wave, XCC id
#define GPROF(i) do { if (threadIdx.x == 0 && blockIdx.x % 49 == 0 && blockIdx.y == 3) g_gather_prof[blockIdx.x / 49][i] = wall_clock64(); \
    if (threadIdx.x == 0 && (i) == 0) g_gather_blk[blockIdx.y][blockIdx.x][0] = wall_clock64(); \
    if (threadIdx.x == 0 && (i) == 5) g_gather_blk[blockIdx.y][blockIdx.x][1] = wall_clock64(); \
    if (threadIdx.x == blockDim.x - 1 && (i) == 5) g_gather_blk[blockIdx.y][blockIdx.x][2] = wall_clock64(); } while (0)
#else
#define GPROF(i) do { } while (0)
#endif

template <typename T, int VT, int MAXL>
__global__ void __launch_bounds__(1024) k_gather_nhwc(GatherArgs a, int tpc)
{
    constexpr int PPL = Lane<T>::PPL;
    extern __shared__ float s_tile[];            // [C][kGTile + 1] fused scores
    __shared__ uint32_t s_best[kGTile];          // max over classes (bits of a float >= 0)
    const int tid = threadIdx.x, b = blockIdx.y;
    GPROF(0);
    const int A = a.t.A, C = a.t.C;
    const int r0 = blockIdx.x * kGTile;
    const int cand = tid / tpc, w = tid - cand * tpc;          // blockDim.x == kGTile * tpc
    if (tid < kGTile) s_best[tid] = 0u;
    // every thread resolves its candidate itself (the threads of a candidate read the same words:
    // one broadcast request): the only dependent memory round trips of the kernel are
    // candidate index -> {class vectors, IoU logit, box deltas}
    const int r_raw = r0 + cand;
    const int r = r_raw < a.R ? r_raw : a.R - 1;
    const LevelSel lv = level_of_candidate<MAXL>(a, r);
    const int idx = a.cand_idx[(size_t)b * a.R + r];
    GPROF(1);
    const int pos = idx / A, an = idx - pos * A;
    const size_t row = ((size_t)b * lv.H * lv.W + pos) * A + an;       // (B, HW, A) row id
    float x[VT][PPL];
#pragma unroll
    for (int u = 0; u < VT; ++u)
        Lane<T>::load_cached(static_cast<const T *>(lv.cls) + row * C + (w + u * tpc) * PPL, x[u]);
    const float iou = load_f32<T>(static_cast<const T *>(lv.iou) + row);
    // the first thread of a candidate also decodes its box (a few lanes of every wavefront,
    // ahead of the barrier, rather than one wavefront's worth behind it)
    const bool boxer = w == 0 && r_raw < a.R;
    float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f, ba0 = 0.0f, ba1 = 0.0f, ba2 = 0.0f, ba3 = 0.0f;
    if (boxer) {
        const T *reg = static_cast<const T *>(lv.reg) + row * 4;
        d0 = load_f32<T>(reg); d1 = load_f32<T>(reg + 1); d2 = load_f32<T>(reg + 2); d3 = load_f32<T>(reg + 3);
        const float *ba = a.ba.v[lv.l][an];
        ba0 = ba[0]; ba1 = ba[1]; ba2 = ba[2]; ba3 = ba[3];
    }
    GPROF(2);
    const float sq = sqrt_sigmoidf_(iou);
    float best = 0.0f;                                          // scores are >= 0
#pragma unroll
    for (int u = 0; u < VT; ++u)
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const float sc = sqrt_sigmoidf_(x[u][j]) * sq;
            s_tile[((w + u * tpc) * PPL + j) * (kGTile + 1) + cand] = sc;
            best = (best < sc) ? sc : best;
        }
    if (boxer)
        reinterpret_cast<float4 *>(a.boxes)[(size_t)b * a.R + r] =
            decode_box(a, b, ba0, ba1, ba2, ba3, lv.W, lv.stride, pos, d0, d1, d2, d3);
    GPROF(3);
    __syncthreads();                                            // s_best cleared, tile written
    atomicMax(&s_best[cand], to_bits(best));
    __syncthreads();
    GPROF(4);
    const int nlive = (a.R - r0 < kGTile) ? (a.R - r0) : kGTile;
    // class-major rows, kGTile contiguous floats each
    float *so = a.scores_t + (size_t)b * C * a.Rs + r0;
    for (int q = tid; q < C * kGTile; q += blockDim.x) {
        const int c = q / kGTile, j = q - c * kGTile;
        if (j < nlive) so[(size_t)c * a.Rs + j] = s_tile[c * (kGTile + 1) + j];
    }
    if (tid < nlive && a.best_score)
        a.best_score[(size_t)b * a.R + r0 + tid] = from_bits(s_best[tid]);
    GPROF(5);
}

// softmax head (see k_rowscore_softmax): one thread per candidate, the C foreground scores from the
// C + 1 logits of its row, class-major like the other gather kernels; both memory orders
template <typename T>
__global__ void __launch_bounds__(64) k_gather_softmax(GatherArgs a)
{
    const int b = blockIdx.y;
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= a.R) return;
    const int A = a.t.A, C = a.t.C, Cin = C + 1;
    const LevelSel lv = level_of_candidate(a, r);
    const int W = lv.W, HW = lv.H * W;
    const int idx = a.cand_idx[(size_t)b * a.R + r];
    const int pos = idx / A, an = idx - pos * A;
    const bool nhwc = a.t.layout == IA_LAYOUT_NHWC;
    const size_t cs = nhwc ? (size_t)1 : (size_t)HW;
    const size_t row = ((size_t)b * HW + pos) * A + an;
    const T *cls = static_cast<const T *>(lv.cls) + (nhwc ? row * Cin : ((size_t)b * A + an) * Cin * HW + pos);
    const T *iou = static_cast<const T *>(lv.iou) + (nhwc ? row : ((size_t)b * A + an) * HW + pos);
    const float sq_iou = sqrt_sigmoidf_(load_f32<T>(iou));
    float m, s, e_fg;
    softmax_row_stats<T>(cls, cs, Cin, m, s, e_fg);
    float *so = a.scores_t + (size_t)b * C * a.Rs + r;
    float best = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float sc = __builtin_sqrtf(expf_(load_f32<T>(cls + (size_t)(c + 1) * cs) - m) / s) * sq_iou;
        so[(size_t)c * a.Rs] = sc;
        best = (best < sc) ? sc : best;
    }
    if (a.best_score) a.best_score[(size_t)b * a.R + r] = best;
    const T *reg = static_cast<const T *>(lv.reg) + (nhwc ? row * 4 : ((size_t)b * A + an) * 4 * HW + pos);
    const float *ba = a.ba.v[lv.l][an];
    reinterpret_cast<float4 *>(a.boxes)[(size_t)b * a.R + r] =
        decode_box(a, b, ba[0], ba[1], ba[2], ba[3], W, lv.stride, pos, load_f32<T>(reg), load_f32<T>(reg + cs),
                   load_f32<T>(reg + 2 * cs), load_f32<T>(reg + 3 * cs));
}

int launch_gather(const LevelTable &t, const BaseAnchors &ba, const float *means, const float *stds,
                  const ia_level_ptrs &p, int batch, int dtype, const int32_t *cand_idx,
                  const float *img_hw, const float *scale_factor, int rescale, float *boxes,
                  float *scores_t, float *best_score, int Rs, hipStream_t s)
{
    if (batch < 1 || !cand_idx || !img_hw || !boxes || !scores_t) return IA_E_ARG;
    if (rescale && !scale_factor) return IA_E_ARG;
    GatherArgs a;
    a.t = t; a.ba = ba; a.p = p;
    for (int i = 0; i < 4; ++i) { a.means[i] = means[i]; a.stds[i] = stds[i]; }
    a.cand_idx = cand_idx; a.img_hw = img_hw; a.scale_factor = scale_factor;
    a.boxes = boxes; a.scores_t = scores_t; a.best_score = best_score;
    a.R = t.cand_off[t.num_levels]; a.Rs = Rs; a.rescale = rescale;
    if (dtype != IA_F32 && dtype != IA_BF16) return IA_E_ARG;
    if (t.softmax) {
        const dim3 grid((unsigned)((a.R + 63) / 64), (unsigned)batch);
        if (dtype == IA_F32) hipLaunchKernelGGL(k_gather_softmax<float>, grid, dim3(64), 0, s, a);
        else hipLaunchKernelGGL(k_gather_softmax<uint16_t>, grid, dim3(64), 0, s, a);
        return hip_status(hipGetLastError());
    }
    const int esz = dtype == IA_F32 ? 4 : 2;
    if (t.layout == IA_LAYOUT_NHWC && (t.C * esz) % 16 == 0 && t.C * esz / 16 <= kMaxVpr) {
        bool aligned = true;
        for (int l = 0; l < t.num_levels; ++l) aligned = aligned && (((uintptr_t)p.cls[l] & 15u) == 0);
        if (aligned) {
            const int vpr = t.C * esz / 16;
            const dim3 grid((unsigned)((a.R + kGTile - 1) / kGTile), (unsigned)batch);
            const size_t lds = (size_t)t.C * (kGTile + 1) * sizeof(float);
            const bool two = vpr % 2 == 0;
            const int tpc = two ? vpr / 2 : vpr;               // threads per candidate
            const dim3 block((unsigned)(kGTile * tpc));
            const bool few = t.num_levels <= 5;                 // the usual P3..P7 pyramid
#define IA_GATHER_LAUNCH(TT, VT)                                                                  \
    do {                                                                                          \
        if (few) hipLaunchKernelGGL((k_gather_nhwc<TT, VT, 5>), grid, block, lds, s, a, tpc);      \
        else hipLaunchKernelGGL((k_gather_nhwc<TT, VT, IA_MAX_LEVELS>), grid, block, lds, s, a, tpc); \
    } while (0)
            if (dtype == IA_F32) {
                if (two) IA_GATHER_LAUNCH(float, 2); else IA_GATHER_LAUNCH(float, 1);
            } else {
                if (two) IA_GATHER_LAUNCH(uint16_t, 2); else IA_GATHER_LAUNCH(uint16_t, 1);
            }
#undef IA_GATHER_LAUNCH
            return hip_status(hipGetLastError());
        }
    }
    dim3 block(64, kGroups);
    dim3 grid((unsigned)((a.R + 63) / 64), (unsigned)batch);
    if (dtype == IA_F32) hipLaunchKernelGGL(k_gather<float>, grid, block, 0, s, a);
    else if (dtype == IA_BF16) hipLaunchKernelGGL(k_gather<uint16_t>, grid, block, 0, s, a);
    else return IA_E_ARG;
    return hip_status(hipGetLastError());
}

}  // namespace ia
