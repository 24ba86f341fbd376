// Per-(image, level) top-k of the row-max scores
// (reference iou_aware_retina_head.py:536-544: `_, topk_inds = max_scores.topk(nms_pre)`).
//
// torch.topk returns indices in descending-score order; the tie order on CPU
// is implementation defined, so the canonical order here is (score
// descending, anchor index ascending) -- identical to the oracle.
//
// Exact selection by threshold.  The k-th largest 32-bit score key T of a segment is found with
// three chip-wide histogram passes over the (L2-resident, 806 KB per image) row-max array --
// digits of 11 / 11 / 10 bits, every workgroup bins 4096 scores in LDS and adds its non-empty
// bins to the segment's global histogram; the pass for digit d+1 only looks at the scores that
// share the threshold's digits 0..d, which every workgroup re-derives from the previous
// histogram.  A fourth pass collects the keys above T (they are selected whatever their order)
// and the indices of the scores equal to T; one workgroup per segment then takes the lowest
// anchor indices among the ties, sorts the k 64-bit keys (score desc, index asc) in LDS and
// writes the candidate list.
//
//   k_sel_hist<0,1,2>   3 launches, ~1600 workgroups each at batch 8
//   k_sel_collect       1 launch
//   k_sel_final         1 workgroup per (image, level)
//
// MI355X, batch 8, 800x1344 (rocprofv3, random-init-like near-tied scores and tie-free scores
// alike): 9.7 + 9.1 + 8.8 + 10.8 + 15.6 = 54 us for the five launches (plus one memset) where
// the earlier two-round radix select (one 1024-thread workgroup per ~9k-score part keeping its
// own top-k, then a merge workgroup per segment: 55 + 53 us) spent its time in the barrier
// chains of 176 workgroups.  Levels with N_l <= nms_pre keep their natural order (the reference
// skips topk there, :537).
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_block.hpp"

namespace ia {

constexpr int kChunk = 4096;               // scores per histogram workgroup
constexpr int kHistThreads = 256;
constexpr int kBins = 2048;                // 11-bit digits (the last digit has 10 bits)
constexpr int kFinalThreads = 1024;
constexpr int kCntStride = 64;             // the two counters of a segment own a 256-byte line pair

struct SelArgs {
    LevelTable t;
    const float *rowmax;
    int32_t *cand_idx;
    uint32_t *hist;                        // (B, L, 3, kBins)
    uint32_t *counters;                    // (B, L, kCntStride): [0] keys above the threshold, [1] ties
    uint64_t *sure;                        // (B, R): keys above the threshold, unordered
    uint32_t *ties;                        // (B, N): anchor indices of scores equal to it
    int32_t chunk_off[IA_MAX_LEVELS + 1];  // prefix of ceil(N_l / kChunk) over selecting levels
    int32_t anchors_per_img, cands_per_img;
};

__device__ __forceinline__ int digit_of(uint32_t key, int level)
{
    return level == 0 ? (int)(key >> 21) : (level == 1 ? (int)((key >> 10) & 0x7ffu) : (int)(key & 0x3ffu));
}

// Threshold digit of one level from its histogram: the bin d (from the top) where the running
// count reaches `need`.  Whole workgroup (256 threads); returns (d, count above d) to everyone.
__device__ __forceinline__ void find_digit(const uint32_t *hist, int nbins, uint32_t need,
                                           uint32_t *lds /* >= 260 */, int &d, uint32_t &above)
{
    const int tid = threadIdx.x;
    const int per = nbins / kHistThreads;          // 8 or 4 bins per thread, highest bins first
    const int hi = nbins - per * tid;
    uint32_t c[8];
    uint32_t s = 0;
    for (int j = 0; j < per; ++j) { c[j] = hist[hi - 1 - j]; s += c[j]; }
    // inclusive scan over the threads (tid 0 owns the top bins)
    uint32_t incl = s;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, off);
        if ((tid & (kWave - 1)) >= off) incl += v;
    }
    const int w = tid >> 6;
    if ((tid & 63) == 63) lds[w] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < w; ++i) base += lds[i];
    incl += base;
    const uint32_t excl = incl - s;
    if (excl < need && incl >= need) {             // exactly one thread
        uint32_t a = excl;
        int dd = hi - 1;
        for (int j = 0; j < per; ++j) {
            if (a + c[j] >= need) { dd = hi - 1 - j; break; }
            a += c[j];
        }
        lds[4] = (uint32_t)dd; lds[5] = a;
    }
    __syncthreads();
    d = (int)lds[4]; above = lds[5];
    __syncthreads();
}

struct SegRef { int l, b; uint32_t n, k, beg, cnt; const float *src; bool natural; uint32_t HW, A; };

__device__ __forceinline__ SegRef locate_chunk(const SelArgs &a)
{
    SegRef r;
    r.b = blockIdx.y;
    int l = 0;
    while ((int)blockIdx.x >= a.chunk_off[l + 1]) ++l;
    r.l = l;
    r.n = (uint32_t)(a.t.anchor_off[l + 1] - a.t.anchor_off[l]);
    r.k = (uint32_t)(a.t.cand_off[l + 1] - a.t.cand_off[l]);
    r.beg = (uint32_t)(blockIdx.x - a.chunk_off[l]) * kChunk;
    r.cnt = (r.n - r.beg < (uint32_t)kChunk) ? (r.n - r.beg) : (uint32_t)kChunk;
    r.src = a.rowmax + (size_t)r.b * a.anchors_per_img + a.t.anchor_off[l];
    r.natural = a.t.layout == IA_LAYOUT_NHWC;
    r.HW = (uint32_t)(a.t.H[l] * a.t.W[l]); r.A = (uint32_t)a.t.A;
    return r;
}

// the thresholds of the levels below `upto` (re-derived by every workgroup from the global
// histograms of the earlier launches): prefix = the digits found so far, need = what is still
// missing among the scores that share them
__device__ __forceinline__ void thresholds(const SelArgs &a, const SegRef &r, int upto, uint32_t *lds,
                                           uint32_t &prefix, uint32_t &need)
{
    const uint32_t *h = a.hist + ((size_t)r.b * a.t.num_levels + r.l) * 3 * kBins;
    prefix = 0; need = r.k;
    for (int lev = 0; lev < upto; ++lev) {
        int d; uint32_t above;
        find_digit(h + lev * kBins, lev == 2 ? 1024 : kBins, need, lds, d, above);
        need -= above;
        prefix = (lev == 2) ? ((prefix << 10) | (uint32_t)d) : ((prefix << 11) | (uint32_t)d);
    }
}

template <int LEVEL>
__global__ void __launch_bounds__(kHistThreads) k_sel_hist(SelArgs a)
{
    __shared__ uint32_t s_hist[kBins];
    __shared__ uint32_t s_misc[264];
    const SegRef r = locate_chunk(a);
    for (int i = threadIdx.x; i < kBins; i += kHistThreads) s_hist[i] = 0;
    uint32_t prefix = 0, need = r.k;
    if (LEVEL > 0) thresholds(a, r, LEVEL, s_misc, prefix, need);
    __syncthreads();
    constexpr int U = kChunk / kHistThreads;       // 16 scores per thread, loaded up front
    uint32_t key[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t j = (uint32_t)u * kHistThreads + threadIdx.x;
        key[u] = (j < r.cnt) ? ordered_key(r.src[r.beg + j]) : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t j = (uint32_t)u * kHistThreads + threadIdx.x;
        bool act = j < r.cnt;
        if (LEVEL == 1) act = act && (key[u] >> 21) == prefix;
        if (LEVEL == 2) act = act && (key[u] >> 10) == prefix;
        hist_add(s_hist, act, (uint32_t)digit_of(key[u], LEVEL));
    }
    __syncthreads();
    uint32_t *g = a.hist + (((size_t)r.b * a.t.num_levels + r.l) * 3 + LEVEL) * kBins;
    for (int i = threadIdx.x; i < kBins; i += kHistThreads)
        if (s_hist[i]) atomicAdd(g + i, s_hist[i]);
}

__global__ void __launch_bounds__(kHistThreads) k_sel_collect(SelArgs a)
{
    __shared__ uint32_t s_misc[264];
    const SegRef r = locate_chunk(a);
    uint32_t T, need;
    thresholds(a, r, 3, s_misc, T, need);          // T: 32-bit key of the k-th largest score
    uint64_t *sure = a.sure + (size_t)r.b * a.cands_per_img + a.t.cand_off[r.l];
    uint32_t *ties = a.ties + (size_t)r.b * a.anchors_per_img + a.t.anchor_off[r.l];
    uint32_t *cnt = a.counters + ((size_t)r.b * a.t.num_levels + r.l) * kCntStride;
    constexpr int U = kChunk / kHistThreads;
    uint32_t key[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t j = (uint32_t)u * kHistThreads + threadIdx.x;
        key[u] = (j < r.cnt) ? ordered_key(r.src[r.beg + j]) : 0u;
    }
    // Slots: every wavefront round reserves its block in LDS counters (returning LDS atomics are
    // cheap), ONE returning global atomic per workgroup and list reserves the workgroup's range.
    // (One global atomic per wavefront round serialised on the few cache lines that hold the
    // counters of all segments: 55 us for this kernel.)
    uint32_t *s_cnt = s_misc + 8;                  // [0] above, [1] ties, [2],[3] global bases
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t off_up[U], off_eq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t j = (uint32_t)u * kHistThreads + threadIdx.x;
        const bool in = j < r.cnt;
        const bool up = in && key[u] > T, eq = in && key[u] == T;
        const uint64_t mu = __ballot(up), me = __ballot(eq);
        off_up[u] = off_eq[u] = 0xffffffffu;
        if (mu) {
            uint32_t b0 = 0;
            const int leader = __builtin_ctzll(mu);
            if (lane_id() == leader) b0 = atomicAdd(&s_cnt[0], (uint32_t)__builtin_popcountll(mu));
            b0 = (uint32_t)__shfl((int)b0, leader);
            if (up) off_up[u] = b0 + lane_prefix_popc(mu);
        }
        if (me) {
            uint32_t b0 = 0;
            const int leader = __builtin_ctzll(me);
            if (lane_id() == leader) b0 = atomicAdd(&s_cnt[1], (uint32_t)__builtin_popcountll(me));
            b0 = (uint32_t)__shfl((int)b0, leader);
            if (eq) off_eq[u] = b0 + lane_prefix_popc(me);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_cnt[threadIdx.x])
        s_cnt[2 + threadIdx.x] = atomicAdd(cnt + threadIdx.x, s_cnt[threadIdx.x]);
    __syncthreads();
    const uint32_t base_up = s_cnt[2], base_eq = s_cnt[3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t j = (uint32_t)u * kHistThreads + threadIdx.x;
        const uint32_t i = r.beg + j;
        // NCHW heads store the row maxima anchor-major (a, p); the reference's anchor index is
        // p*A + a, which is the storage order itself for channels-last heads
        const uint32_t an = i / r.HW, p = i - an * r.HW;
        const uint32_t idx = r.natural ? i : (p * r.A + an);
        if (off_up[u] != 0xffffffffu)
            sure[base_up + off_up[u]] = ((uint64_t)key[u] << 32) | (uint64_t)(0xffffffffu - idx);
        if (off_eq[u] != 0xffffffffu) ties[base_eq + off_eq[u]] = idx;
    }
    if (blockIdx.x == (unsigned)a.chunk_off[r.l] && threadIdx.x == 0) {
        // leave the threshold for the final kernel (one writer per segment)
        a.hist[(((size_t)r.b * a.t.num_levels + r.l) * 3 + 2) * kBins + 1024] = T;
        a.hist[(((size_t)r.b * a.t.num_levels + r.l) * 3 + 2) * kBins + 1025] = need;
    }
}

__global__ void __launch_bounds__(kFinalThreads) k_sel_final(SelArgs a)
{
    __shared__ TopkScratch sc;
    __shared__ uint64_t sel[IA_MAX_NMS_PRE];
    const int l = blockIdx.x, b = blockIdx.y;
    const uint32_t n = (uint32_t)(a.t.anchor_off[l + 1] - a.t.anchor_off[l]);
    const uint32_t k = (uint32_t)(a.t.cand_off[l + 1] - a.t.cand_off[l]);
    int32_t *out = a.cand_idx + (size_t)b * a.cands_per_img + a.t.cand_off[l];
    if (k == n) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) out[i] = (int32_t)i;
        return;
    }
    const uint32_t *h2 = a.hist + (((size_t)b * a.t.num_levels + l) * 3 + 2) * kBins;
    const uint32_t T = h2[1024], need = h2[1025];       // need = ties to take, 1 <= need <= n_tie
    const uint32_t *cnt = a.counters + ((size_t)b * a.t.num_levels + l) * kCntStride;
    const uint32_t n_sure = cnt[0], n_tie = cnt[1];     // n_sure + need == k
    const uint64_t *sure = a.sure + (size_t)b * a.cands_per_img + a.t.cand_off[l];
    const uint32_t *ties = a.ties + (size_t)b * a.anchors_per_img + a.t.anchor_off[l];
    const uint32_t P = next_pow2(k);
    // the ties that are taken: the `need` lowest anchor indices among the scores equal to T
    // (usually need == n_tie == 1).  sel doubles as the selection's output buffer; the chosen
    // keys wait in registers while sel is refilled with the keys above the threshold.
    uint64_t mine[IA_MAX_NMS_PRE / kFinalThreads];
    if (n_tie != need) {
        block_topk_desc([ties, T](uint32_t i) -> uint64_t {
                            return ((uint64_t)T << 32) | (uint64_t)(0xffffffffu - ties[i]); },
                        n_tie, need, sc, sel);
    }
#pragma unroll
    for (int u = 0; u < IA_MAX_NMS_PRE / kFinalThreads; ++u) {
        const uint32_t i = (uint32_t)u * kFinalThreads + threadIdx.x;
        mine[u] = 0ull;
        if (i < need)
            mine[u] = (n_tie != need) ? sel[i]
                                      : (((uint64_t)T << 32) | (uint64_t)(0xffffffffu - ties[i]));
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) sel[i] = (i < n_sure) ? sure[i] : 0ull;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < IA_MAX_NMS_PRE / kFinalThreads; ++u) {
        const uint32_t i = (uint32_t)u * kFinalThreads + threadIdx.x;
        if (i < need) sel[n_sure + i] = mine[u];
    }
    __syncthreads();
    bitonic_sort_desc(sel, P);
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x)
        out[i] = (int32_t)(0xffffffffu - (uint32_t)sel[i]);
}

static void plan(const LevelTable &t, SelArgs &a)
{
    a.chunk_off[0] = 0;
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        int chunks = 0;
        if (l < t.num_levels) {
            const int n = t.anchor_off[l + 1] - t.anchor_off[l];
            const int k = t.cand_off[l + 1] - t.cand_off[l];
            if (k < n) chunks = (n + kChunk - 1) / kChunk;
        }
        a.chunk_off[l + 1] = a.chunk_off[l] + chunks;
    }
}

struct SelLayout { size_t hist, counters, sure, ties, total; };

static SelLayout layout(const LevelTable &t, int batch)
{
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    SelLayout w;
    const size_t L = (size_t)t.num_levels, B = (size_t)batch;
    size_t o = 0;
    w.hist = o; o = up(o + B * L * 3 * kBins * sizeof(uint32_t));
    w.counters = o; o = up(o + B * L * kCntStride * sizeof(uint32_t));
    const size_t zeroed = o;                       // hist + counters are cleared per call
    w.sure = o; o = up(o + B * (size_t)t.cand_off[t.num_levels] * sizeof(uint64_t));
    w.ties = o; o = up(o + B * (size_t)t.anchor_off[t.num_levels] * sizeof(uint32_t));
    w.total = o;
    (void)zeroed;
    return w;
}

size_t select_workspace_bytes(const LevelTable &t, int batch)
{
    return layout(t, batch).total;
}

int launch_select(const LevelTable &t, const float *rowmax, int batch, int32_t *cand_idx,
                  void *workspace, hipStream_t s)
{
    if (batch < 1 || !rowmax || !cand_idx || !workspace) return IA_E_ARG;
    SelArgs a;
    a.t = t; a.rowmax = rowmax; a.cand_idx = cand_idx;
    const SelLayout w = layout(t, batch);
    char *ws = static_cast<char *>(workspace);
    a.hist = reinterpret_cast<uint32_t *>(ws + w.hist);
    a.counters = reinterpret_cast<uint32_t *>(ws + w.counters);
    a.sure = reinterpret_cast<uint64_t *>(ws + w.sure);
    a.ties = reinterpret_cast<uint32_t *>(ws + w.ties);
    a.anchors_per_img = t.anchor_off[t.num_levels];
    a.cands_per_img = t.cand_off[t.num_levels];
    plan(t, a);
    const int chunks = a.chunk_off[t.num_levels];
    if (chunks > 0) {
        hipError_t e = hipMemsetAsync(ws + w.hist, 0, w.sure - w.hist, s);
        if (e != hipSuccess) return (int)e;
        const dim3 grid((unsigned)chunks, (unsigned)batch), block(kHistThreads);
        hipLaunchKernelGGL(k_sel_hist<0>, grid, block, 0, s, a);
        hipLaunchKernelGGL(k_sel_hist<1>, grid, block, 0, s, a);
        hipLaunchKernelGGL(k_sel_hist<2>, grid, block, 0, s, a);
        hipLaunchKernelGGL(k_sel_collect, grid, block, 0, s, a);
        int rc = hip_status(hipGetLastError());
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_sel_final, dim3((unsigned)t.num_levels, (unsigned)batch),
                       dim3(kFinalThreads), 0, s, a);
    return hip_status(hipGetLastError());
}

}  // namespace ia
