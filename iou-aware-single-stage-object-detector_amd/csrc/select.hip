// Per-(image, level) top-k of the row-max scores
// (reference iou_aware_retina_head.py:536-544: `_, topk_inds = max_scores.topk(nms_pre)`).
//
// torch.topk returns indices in descending-score order; the tie order on CPU
// is implementation defined, so the canonical order here is (score
// descending, anchor index ascending) -- identical to the oracle.
//
// Two launches behind the row-max kernel (before: three chip-wide histogram passes + a collect
// pass + a one-workgroup sort, 5 launches and 72-90 us at batch 8):
//
//   k_sel_filter   large levels (N_l > kSelDenseMax).  The row-max kernel leaves, next to the
//                  scores, the maximum of every group of g = 64 / 16 / 4 consecutive stored scores
//                  (ia_internal.hpp, SelPlan).  With v <= the k-th largest group maximum of a
//                  segment, k distinct scores are >= v, hence the k-th largest score T >= v and
//                  every member of the top-k is >= v.  Each workgroup derives v from <= 4096 group
//                  maxima -- ONE histogram pass over 2048 linear bins between their minimum and
//                  maximum, v = the lower edge of the bin where the count from the top reaches k
//                  -- and keeps the scores >= v of its 4096-score chunk as 64-bit keys
//                  `ordered(score) << 32 | ~index` in the chunk's own slice of the candidate list
//                  (no global atomics, nothing to clear).  Independent scores leave about 1.3 k
//                  candidates of 151 200; the bound is valid for ANY input (all scores equal:
//                  every score is a candidate and the final kernel takes its general path).
//   k_sel_final    one workgroup per segment: the candidates (or the whole level when it is
//                  small) are staged in LDS as unique 64-bit keys; one histogram pass over linear
//                  bins between the smallest and largest staged score finds the bin holding the
//                  cut; the keys above it are selected outright, the few inside it are ranked
//                  against each other; bitonic sort with the short strides in registers
//                  (ia_block.hpp) -> canonical order.  Inputs the shortcut does not fit (more than
//                  1024 keys in the cut's bin -- heavy ties -- or more candidates than the LDS
//                  stage holds) take the exact MSB-first radix select of ia_block.hpp.
//
// Levels with N_l <= nms_pre keep their natural order (the reference skips topk there, :537).
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_rowmax_dev.hpp"

// phase timestamps for tools/ubench/select_bench.hip (compiled out of the library)
#ifdef IA_SEL_PROFILE
namespace ia { __device__ unsigned long long g_sel_prof[2][IA_MAX_LEVELS][24];
               __device__ unsigned long long g_sel_blk[2][16][64][2];     // per-block start / end
               __device__ unsigned long long g_blk_t[8192][3]; }          // fused launch, per workgroup: start, end, wait end
#define SEL_PROF(kern, lvl, i)                                                        \
    do { if (blockIdx.y == 0 && threadIdx.x == 0) ia::g_sel_prof[kern][lvl][i] = wall_clock64(); } while (0)
#define IA_BLOCK_PROF(i) SEL_PROF(1, blockIdx.x, i)
#else
#define SEL_PROF(kern, lvl, i) do { } while (0)
#endif
#include "ia_block.hpp"

namespace ia {

constexpr int kFilterThreads = 256;
constexpr int kBins = 2048;
constexpr int kFinalThreads = 1024;
constexpr int kMaxGroups = 4096;           // group maxima a filter workgroup looks at (2^12)
constexpr int kGroupsPerThread = kMaxGroups / kFilterThreads;
constexpr int kMaxSegChunks = 1024;        // filter chunks per segment (N_l <= 4 M anchors)

struct SelArgs {
    LevelTable t;
    SelPlan plan;
    const float *rowmax;
    const uint32_t *groupmax;              // group maxima as ordered keys (never 0)
    int32_t *cand_idx;
    uint64_t *seg_v;                       // (B, L): {1, threshold key} once a segment's leader has it
    uint32_t *chunk_count;                 // (B, chunks): candidates each filter workgroup kept
    uint64_t *cand;                        // (B, N): candidate keys; chunk c of segment (b, l) owns
                                           // [b * N + anchor_off[l] + c * kSelChunk, + kSelChunk)
    int32_t batch, anchors_per_img, cands_per_img, lds_cap, total_chunks;
    // fused launch only (0 otherwise): the id of this call -- what a filter workgroup that gives up
    // waiting writes into the status word, and what k_sel_final compares the word with (a stale
    // word of an earlier call never matches: nothing has to clear it) -- and the spin bound
    uint32_t call_id, spin_limit;
};

// The bin d (from the top) where the running count of a 2048-bin histogram reaches `need`, by a
// 256-thread workgroup; returns (d, count above d, count in d) to everyone.
__device__ __forceinline__ void find_bin_256(const uint32_t *hist, uint32_t need,
                                             uint32_t *lds /* >= 8 */, int &d, uint32_t &above,
                                             uint32_t &in_d)
{
    const int tid = threadIdx.x;
    constexpr int per = kBins / kFilterThreads;    // 8 bins per thread, highest bins first
    const int hi = kBins - per * tid;
    uint32_t c[per];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < per; ++j) { c[j] = hist[hi - 1 - j]; s += c[j]; }
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
        uint32_t a = excl, cc = c[0];
        int dd = hi - 1;
#pragma unroll
        for (int j = 0; j < per; ++j) {
            if (a + c[j] >= need) { dd = hi - 1 - j; cc = c[j]; break; }
            a += c[j];
        }
        lds[4] = (uint32_t)dd; lds[5] = a; lds[6] = cc;
    }
    __syncthreads();
    d = (int)lds[4]; above = lds[5]; in_d = lds[6];
}

// shift that maps [0, range] onto at most kBins linear bins
__device__ __forceinline__ int bin_shift(uint32_t range)
{
    const int bits = 32 - __builtin_clz(range | 1u);
    return bits > 11 ? bits - 11 : 0;
}

struct SegRef { int l, b; uint32_t n, k, beg, cnt, chunk; const float *src; bool natural; uint32_t HW, A; };

template <typename AT>
__device__ __forceinline__ SegRef locate_chunk(const AT &a, int cx, int b)
{
    SegRef r;
    r.b = b;
    int l = 0;
    while (cx >= a.plan.chunk_off[l + 1]) ++l;
    r.l = l;
    r.n = (uint32_t)(a.t.anchor_off[l + 1] - a.t.anchor_off[l]);
    r.k = (uint32_t)(a.t.cand_off[l + 1] - a.t.cand_off[l]);
    r.chunk = (uint32_t)(cx - a.plan.chunk_off[l]);
    r.beg = r.chunk * kSelChunk;
    r.cnt = (r.n - r.beg < (uint32_t)kSelChunk) ? (r.n - r.beg) : (uint32_t)kSelChunk;
    r.src = a.rowmax + (size_t)r.b * a.anchors_per_img + a.t.anchor_off[l];
    r.natural = a.t.layout == IA_LAYOUT_NHWC;
    r.HW = (uint32_t)(a.t.H[l] * a.t.W[l]); r.A = (uint32_t)a.t.A;
    return r;
}

// the groups that lie completely inside segment (b, l): [first, first + count) of the level's array
// (g is 4, 16 or 64: shifts, no 64-bit divisions)
__device__ __forceinline__ void segment_groups(const LevelTable &t, int l, int b, int g,
                                               int64_t &first, int64_t &count)
{
    const int lg = 31 - __builtin_clz((unsigned)g);
    const int64_t n = t.anchor_off[l + 1] - t.anchor_off[l];
    if (t.layout == IA_LAYOUT_NHWC) {
        count = (n + g - 1) >> lg;                       // per-image group words (the last one may be partial)
        first = (int64_t)b * count;
    } else {
        const int64_t gpp = ((int64_t)t.H[l] * t.W[l] + g - 1) >> lg;
        first = (int64_t)b * t.A * gpp;
        count = (int64_t)t.A * gpp;
    }
}

// Fallback producer of the group maxima (the stage-wise C-ABI, where the row-max array comes from
// the caller): one thread per group over the stored scores.
__global__ void __launch_bounds__(256) k_sel_groupmax(SelArgs a, uint32_t *groupmax)
{
    const int64_t gid0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int l = 0;
    while (l < a.t.num_levels && gid0 >= a.plan.goff[l + 1]) ++l;
    if (l >= a.t.num_levels) return;
    const int g = a.plan.grp[l];
    const int64_t gid = gid0 - a.plan.goff[l];
    const int64_t n = a.t.anchor_off[l + 1] - a.t.anchor_off[l];
    float m = 0.0f;                                 // scores are >= 0
    if (a.t.layout == IA_LAYOUT_NHWC) {
        const int64_t gpi = (n + g - 1) / g;             // group words per image
        const int64_t b = gid / gpi, q = gid - b * gpi;
        for (int j = 0; j < g; ++j) {
            const int64_t i = q * g + j;
            if (i >= n) break;
            const float v = a.rowmax[(size_t)b * a.anchors_per_img + a.t.anchor_off[l] + i];
            m = (m < v) ? v : m;
        }
    } else {
        const int64_t hw = (int64_t)a.t.H[l] * a.t.W[l];
        const int64_t gpp = (hw + g - 1) / g;
        const int64_t plane = gid / gpp, q = gid - plane * gpp;      // plane = b * A + an
        const int64_t b = plane / a.t.A, an = plane - b * a.t.A;
        const float *src = a.rowmax + (size_t)b * a.anchors_per_img + a.t.anchor_off[l] + an * hw;
        for (int j = 0; j < g; ++j) {
            const int64_t pos = q * g + j;
            if (pos >= hw) break;
            m = (m < src[pos]) ? src[pos] : m;
        }
    }
    groupmax[gid0] = ordered_key(m);                 // m >= +0: bits | 0x80000000
}

// v from the group keys a workgroup holds in registers (thread tid: keys u * 256 + tid < used).
// A key of 0 is a group whose maximum has not been published yet (real keys have the top bit set):
// it is left out, which can only LOWER v -- the k-th largest of a subset of the group maxima is
// still a lower bound of the segment's k-th largest score.  `present` (block-uniform) = the keys
// that are not 0; fewer than k of them: no filtering (v = 0).  s_hist zero on entry.
__device__ __forceinline__ uint32_t threshold_from_keys(const uint32_t (&gk)[kGroupsPerThread], uint32_t used,
                                                        uint32_t present, uint32_t k, uint32_t *s_hist,
                                                        uint32_t *s_misc)
{
    const int tid = threadIdx.x;
    if (present < k) {
        __syncthreads();
        return 0u;
    }
    uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
    for (int u = 0; u < kGroupsPerThread; ++u) {
        const uint32_t j = (uint32_t)u * kFilterThreads + tid;
        if (j < used && gk[u] != 0u) { lo = gk[u] < lo ? gk[u] : lo; hi = gk[u] > hi ? gk[u] : hi; }
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, off), h2 = (uint32_t)__shfl_xor((int)hi, off);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((tid & 63) == 0) { s_misc[8 + (tid >> 6)] = lo; s_misc[12 + (tid >> 6)] = hi; }
    __syncthreads();                                     // also: s_hist cleared
#pragma unroll
    for (int w = 0; w < kFilterThreads / kWave; ++w) {
        lo = s_misc[8 + w] < lo ? s_misc[8 + w] : lo;
        hi = s_misc[12 + w] > hi ? s_misc[12 + w] : hi;
    }
    const int shift = bin_shift(hi - lo);
#pragma unroll
    for (int u = 0; u < kGroupsPerThread; ++u) {
        const uint32_t j = (uint32_t)u * kFilterThreads + tid;
        if (j < used && gk[u] != 0u) atomicAdd(&s_hist[(gk[u] - lo) >> shift], 1u);
    }
    __syncthreads();
    int d; uint32_t above, in_d;
    find_bin_256(s_hist, k, s_misc, d, above, in_d);
    return lo + ((uint32_t)d << shift);                  // >= k group maxima are >= v
}

// v: a lower bound of segment r's k-th largest key from (a sample of) its group maxima -- ONE
// histogram pass over 2048 linear bins between their minimum and maximum, v = the lower edge of the
// bin where the count from the top reaches k.  s_hist must be zero on entry (and a barrier later).
template <typename AT>
__device__ __forceinline__ uint32_t segment_threshold(const AT &a, const SegRef &r, uint32_t *s_hist,
                                                      uint32_t *s_misc)
{
    const int tid = threadIdx.x;
    const int g = a.plan.grp[r.l];
    int64_t first, count;
    segment_groups(a.t, r.l, r.b, g, first, count);
    const uint32_t stride = (uint32_t)((count + kMaxGroups - 1) >> 12);        // kMaxGroups = 4096
    const uint32_t used = count > 0 ? ((uint32_t)count + stride - 1) / stride : 0u;
    if (used < r.k) {                                    // uniform: too few groups, no filtering
        __syncthreads();
        return 0u;
    }
    const uint32_t *gm = a.groupmax + a.plan.goff[r.l] + first;
    uint32_t gk[kGroupsPerThread];
#pragma unroll
    for (int u = 0; u < kGroupsPerThread; ++u) {
        const uint32_t j = (uint32_t)u * kFilterThreads + tid;
        gk[u] = gm[(size_t)(j < used ? j : used - 1) * stride];    // never predicated
    }
    return threshold_from_keys(gk, used, used, r.k, s_hist, s_misc);
}

// One filter workgroup: the scores >= v of its chunk -> the chunk's slice of the candidate list.
// have_v: the threshold is given (a follower of the fused launch); else derived here.
template <typename AT>
__device__ __forceinline__ void sel_filter_body(const AT &a, const SegRef &r, uint32_t *s_hist,
                                                uint32_t *s_misc /* 24 words */, bool have_v = false,
                                                uint32_t v_given = 0)
{
    const int tid = threadIdx.x;
#ifdef IA_SEL_PROFILE
    const bool prof = r.chunk == 0 && r.b == 0;
#define FPROF(i) do { if (prof) SEL_PROF(0, r.l, i); } while (0)
#else
#define FPROF(i) do { } while (0)
#endif
    FPROF(0);
#ifdef IA_SEL_PROFILE
    const int pcx = a.plan.chunk_off[r.l] + (int)r.chunk;
    if (tid == 0 && r.b < 16 && pcx < 64) g_sel_blk[0][r.b][pcx][0] = wall_clock64();
#endif
    // this chunk's scores: requested first, their latency hides behind the threshold search
    constexpr int U = kSelChunk / kFilterThreads;       // 16 scores per thread
    // (loads are never predicated -- out-of-range lanes read a clamped address: a predicated load
    // compiles to branch + load + s_waitcnt, i.e. 16 serialised round trips)
    float kf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t j = (uint32_t)u * kFilterThreads + tid;
        kf[u] = r.src[r.beg + (j < r.cnt ? j : r.cnt - 1)];
    }
    // ---- v: a lower bound of the segment's k-th largest key from (a sample of) its group maxima
    for (int i = tid; i < kBins; i += kFilterThreads) s_hist[i] = 0;
    if (tid == 0) s_misc[16] = 0;                        // candidates of this chunk
    uint32_t v = 0;
    if (have_v) {
        v = v_given;
        __syncthreads();
    } else {
        v = segment_threshold(a, r, s_hist, s_misc);
    }
    FPROF(3);
    uint32_t key[U];
#pragma unroll
    for (int u = 0; u < U; ++u) key[u] = ordered_key(kf[u]);
    // ---- candidates of this chunk, into the chunk's own slice of the list
    uint64_t *list = a.cand + (size_t)r.b * a.anchors_per_img + a.t.anchor_off[r.l] + r.beg;
    uint32_t hits = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t j = (uint32_t)u * kFilterThreads + tid;
        hits += ((j < r.cnt) && key[u] >= v) ? 1u : 0u;
    }
    uint32_t pos = hits ? atomicAdd(&s_misc[16], hits) : 0u;
    FPROF(4);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t j = (uint32_t)u * kFilterThreads + tid;
        if ((j < r.cnt) && key[u] >= v) {
            const uint32_t i = r.beg + j;
            // NCHW heads store the row maxima anchor-major (a, p); the reference's anchor index is
            // p*A + a, which is the storage order itself for channels-last heads
            const uint32_t an = i / r.HW, p = i - an * r.HW;
            const uint32_t idx = r.natural ? i : (p * r.A + an);
            list[pos++] = ((uint64_t)key[u] << 32) | (uint64_t)(0xffffffffu - idx);
        }
    }
    __syncthreads();
    if (tid == 0)
        a.chunk_count[(size_t)r.b * a.total_chunks + a.plan.chunk_off[r.l] + r.chunk] = s_misc[16];
    FPROF(5);
#ifdef IA_SEL_PROFILE
    if (tid == 0 && r.b < 16 && pcx < 64) g_sel_blk[0][r.b][pcx][1] = wall_clock64();
#endif
}

__global__ void __launch_bounds__(kFilterThreads) k_sel_filter(SelArgs a)
{
    __shared__ uint32_t s_hist[kBins];
    __shared__ uint32_t s_misc[24];
    sel_filter_body(a, locate_chunk(a, (int)blockIdx.x, (int)blockIdx.y), s_hist, s_misc);
}

// ---------------------------------------------------------------------------------------------
// Row-max AND filter in ONE launch (channels-last heads; ia_get_bboxes / ia_decode_stage).
// Workgroup order: per level, first its row-max workgroups (four wavefronts of 64 rows each,
// ia_rowmax_dev.hpp, PUBLISH form), then -- for a filtered level -- its filter workgroups, so
// that a level's filtering runs under the NEXT levels' streaming and the dependent launch with
// its ramp (8-10 us) leaves the critical path.
//   producer  a row-max wavefront writes its scores with write-through stores, drains its memory
//             counter, and only then stores its group maxima: non-zero words (ordered keys);
//   leader    the filter workgroup of a segment's chunk 0 re-reads the segment's group words
//             (agent-scope relaxed loads) until none is zero -- every 64-row unit of the segment
//             is then in memory --, does ONE agent acquire, derives the threshold v and publishes
//             the granule {1, v} for the segment;
//   follower  the other chunks' workgroups poll that one granule (one lane), acquire, filter.
// No atomics, no counters (2 362 wavefronts arriving on one device-scope counter serialised into
// 120 us).  State: the group words and granules must be ZERO when the launch starts; k_sel_final,
// always launched behind this kernel, clears them again (include/iouaware.h: WORKSPACE CONTRACT).
// Residency: the filter workgroups (chunks x batch, 4 wavefronts each) can never fill the chip,
// so a row-max workgroup always finds a slot whatever the dispatch order: no deadlock.  Spins are
// bounded; a timeout leaves a non-zero status word in the workspace.
// SAFETY: a workgroup that gives up waiting stores the call's id in
// the status word and goes on (its candidate slice is then garbage, in bounds).  k_sel_final --
// a later launch on the same stream, so every row maximum IS in memory by then -- sees the id and
// selects from the row maxima of the whole level instead of the candidate lists: the result is the
// same detections, only slower; status[1] counts such calls (telemetry, ops.get_bboxes_status).
constexpr uint32_t kSpinLimit = 1u << 21;
static std::atomic<uint32_t> g_spin_limit{kSpinLimit};   // ia_debug_fused_spin_limit (tests)
static std::atomic<uint32_t> g_fused_calls{0};

// Four / eight L1-bypassing loads in flight per lane (agent-scope atomic loads are issued one at a
// time, each behind a wait: a poll round of 10 words per lane took 10-20 us).
__device__ __forceinline__ void load8_sc1(const uint32_t *const (&q)[8], uint32_t (&w)[8])
{
    asm volatile("global_load_dword %0, %8, off sc1\n\t"
                 "global_load_dword %1, %9, off sc1\n\t"
                 "global_load_dword %2, %10, off sc1\n\t"
                 "global_load_dword %3, %11, off sc1\n\t"
                 "global_load_dword %4, %12, off sc1\n\t"
                 "global_load_dword %5, %13, off sc1\n\t"
                 "global_load_dword %6, %14, off sc1\n\t"
                 "global_load_dword %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                 : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7])
                 : "memory");
}

// Leader of a fused-launch segment: polls the (sampled) group words of the segment until all but
// an eighth of them are published, and derives v from the keys it then holds -- no second read,
// and the threshold is out long before the segment's last row-max wavefront finishes (what a
// follower waits for is its OWN chunk, chunk_ready below).  s_hist zero on entry.
template <typename AT>
__device__ __forceinline__ uint32_t segment_threshold_polled(const AT &a, const SegRef &r, uint32_t *s_hist,
                                                             uint32_t *s_misc, uint32_t *status)
{
    const int tid = threadIdx.x;
    const int g = a.plan.grp[r.l];
    int64_t first, count;
    segment_groups(a.t, r.l, r.b, g, first, count);
    const uint32_t stride = (uint32_t)((count + kMaxGroups - 1) >> 12);
    const uint32_t used = count > 0 ? ((uint32_t)count + stride - 1) / stride : 0u;
    if (used < r.k) {                                    // uniform: too few groups, no filtering
        __syncthreads();
        return 0u;
    }
    uint32_t need = used - (used >> 3);
    need = need < r.k ? r.k : need;
    const uint32_t *gm = a.groupmax + a.plan.goff[r.l] + first;
    uint32_t gk[kGroupsPerThread];
    uint32_t present = 0;
    for (uint32_t spins = 0;; ++spins) {
        uint32_t mine = 0;
#pragma unroll
        for (int u0 = 0; u0 < kGroupsPerThread; u0 += 8) {
            const uint32_t *q[8];
            uint32_t w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t j = (uint32_t)(u0 + u) * kFilterThreads + tid;
                q[u] = gm + (size_t)(j < used ? j : used - 1) * stride;
            }
            load8_sc1(q, w);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t j = (uint32_t)(u0 + u) * kFilterThreads + tid;
                gk[u0 + u] = w[u];
                mine += (j < used && w[u] != 0u) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) mine += (uint32_t)__shfl_xor((int)mine, off);
        __syncthreads();                                 // the previous round's readers are done
        if ((tid & 63) == 0) s_misc[8 + (tid >> 6)] = mine;
        __syncthreads();
        present = (s_misc[8] + s_misc[9]) + (s_misc[10] + s_misc[11]);
        if (present >= need) break;
        if (spins >= a.spin_limit) { if (tid == 0) *status = a.call_id; break; }          // uniform
        __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();                                     // s_misc[8..11] free again
    return threshold_from_keys(gk, used, present, r.k, s_hist, s_misc);
}

// every row of chunk r is in memory: the group words that cover it are published
template <typename AT>
__device__ __forceinline__ void chunk_ready(const AT &a, const SegRef &r, uint32_t *status)
{
    const int tid = threadIdx.x;
    const int lg = 31 - __builtin_clz((unsigned)a.plan.grp[r.l]);
    const int64_t gpi = ((int64_t)r.n + a.plan.grp[r.l] - 1) >> lg;     // group words per image
    const int64_t w0 = (int64_t)r.b * gpi + (r.beg >> lg), w1 = (int64_t)r.b * gpi + ((r.beg + r.cnt - 1) >> lg) + 1;
    const uint32_t *gw = a.groupmax + a.plan.goff[r.l];
    for (uint32_t spins = 0;; ++spins) {
        int ok = 1;
        for (int64_t j = w0 + tid; j < w1; j += kFilterThreads * 8) {
            const uint32_t *q[8];
            uint32_t w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t ju = j + (int64_t)u * kFilterThreads;
                q[u] = gw + (ju < w1 ? ju : w1 - 1);
            }
            load8_sc1(q, w);
#pragma unroll
            for (int u = 0; u < 8; ++u) ok &= (w[u] != 0u);
        }
        if (__syncthreads_and(ok)) break;
        if (spins >= a.spin_limit) { if (tid == 0) *status = a.call_id; break; }          // uniform
        __builtin_amdgcn_s_sleep(4);
    }
}

struct FusedOrder {
    int32_t item_off[3 * IA_MAX_LEVELS + 1];    // prefix of workgroups over the items, in launch order
    int32_t item_level[3 * IA_MAX_LEVELS];      // item -> level
    int32_t item_filter[3 * IA_MAX_LEVELS];     // item -> 0: row-max workgroups, 1: filter leaders
                                                // (chunk 0 of every image), 2: the other filter chunks
    int32_t units[IA_MAX_LEVELS];               // 64-row units of level l
    int32_t n_items;
};

template <typename T, int VPR_T>
__global__ void __launch_bounds__(kFilterThreads, 4) k_rowmax_filter_nhwc(RowmaxNhwcArgs ra, SelArgs sa,
                                                                       FusedOrder fo)
{
    constexpr int kTile = 64 * ((VPR_T ? VPR_T : kMaxVpr) + 1);
    constexpr int kLdsWords = (4 * kTile > kBins + 24) ? 4 * kTile : (kBins + 24);
    __shared__ uint32_t s_raw[kLdsWords];
    int it = 0;
    while ((int)blockIdx.x >= fo.item_off[it + 1]) ++it;
    const int l = fo.item_level[it], local = (int)blockIdx.x - fo.item_off[it];
    const int kind = fo.item_filter[it];
    if (!kind) {
        const int wv = threadIdx.x >> 6;
        const int unit = local * 4 + wv;
#ifdef IA_SEL_PROFILE
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_blk_t[blockIdx.x][0] = wall_clock64();
#endif
        if (unit < fo.units[l])
            rowmax_nhwc_wave<T, VPR_T, true>(ra, l, unit, reinterpret_cast<float *>(s_raw) + wv * kTile,
                                             (int)(threadIdx.x & 63));
#ifdef IA_SEL_PROFILE
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_blk_t[blockIdx.x][1] = wall_clock64();
#endif
        return;
    }
#ifdef IA_ABL_NO_FILTER                                      /* tools/ubench/stage_bench.hip only */
    return;
#endif
    const int nch = sa.plan.chunk_off[l + 1] - sa.plan.chunk_off[l];
    int b, cx;
    if (kind == 1) { b = local; cx = sa.plan.chunk_off[l]; }
    else { b = local / (nch - 1); cx = sa.plan.chunk_off[l] + 1 + (local - b * (nch - 1)); }
    const SegRef r = locate_chunk(sa, cx, b);
    uint32_t *s_hist = s_raw, *s_misc = s_raw + kBins;
    uint64_t *granule = sa.seg_v + (size_t)b * sa.t.num_levels + l;
    uint32_t *status = sa.chunk_count + (size_t)sa.batch * sa.total_chunks;
    const int tid = threadIdx.x;
#ifdef IA_SEL_PROFILE
    if (tid == 0 && blockIdx.x < 8192) g_blk_t[blockIdx.x][0] = wall_clock64();
#endif
    uint32_t v;
    if (r.chunk == 0) {
        // leader: the threshold from the group maxima published so far, then a follower like the rest
        for (int i = tid; i < kBins; i += kFilterThreads) s_hist[i] = 0;
        __syncthreads();
        v = segment_threshold_polled(sa, r, s_hist, s_misc, status);
        if (tid == 0)
            __hip_atomic_store(granule, (1ull << 32) | (uint64_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        chunk_ready(sa, r, status);
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();                                    // s_hist / s_misc are reused below
    } else {
        chunk_ready(sa, r, status);
        if (tid == 0) {
            uint64_t x;
            uint32_t spins = 0;
            while (((x = __hip_atomic_load(granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) == 0) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > sa.spin_limit) { *status = sa.call_id; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_misc[20] = (uint32_t)x;
        }
        __syncthreads();
        v = s_misc[20];
        __syncthreads();
    }
#ifdef IA_SEL_PROFILE
    if (tid == 0 && blockIdx.x < 8192) g_blk_t[blockIdx.x][2] = wall_clock64();
#endif
    sel_filter_body(sa, r, s_hist, s_misc, true, v);
#ifdef IA_SEL_PROFILE
    if (tid == 0 && blockIdx.x < 8192) g_blk_t[blockIdx.x][1] = wall_clock64();
#endif
}

extern __shared__ uint64_t s_dyn[];            // k_sel_final: sel / buckets | staged candidate keys

constexpr int kBucketCap = 128;            // keys per histogram bin the counting sort accepts

__global__ void __launch_bounds__(kFinalThreads) k_sel_final(SelArgs a, uint32_t p_max)
{
    __shared__ TopkScratch sc;                     // sc.hist: bin counts, then bucket fill counters
    __shared__ uint32_t s_start[kBins + 1];        // keys in the bins above bin i
    __shared__ uint32_t s_pre[kMaxSegChunks + 1];
    __shared__ uint32_t s_red[40];
    uint64_t *sel = s_dyn;                         // p_max + kBucketCap entries
    uint64_t *stage = s_dyn + p_max + kBucketCap;
    const int l = blockIdx.x, b = blockIdx.y;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    SEL_PROF(1, l, 0);

    const uint32_t n = (uint32_t)(a.t.anchor_off[l + 1] - a.t.anchor_off[l]);
    const uint32_t k = (uint32_t)(a.t.cand_off[l + 1] - a.t.cand_off[l]);
    int32_t *out = a.cand_idx + (size_t)b * a.cands_per_img + a.t.cand_off[l];
    // a filter workgroup of THIS call timed out: its candidate slice cannot be trusted; the row
    // maxima are complete (kernel boundary), so the level is selected from them like a dense one
    uint32_t *status = a.chunk_count + (size_t)a.batch * a.total_chunks;
    const bool any_timeout = a.call_id != 0u &&
                             __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.call_id;
    // telemetry (status[1] = calls that fell back): counted by the first workgroup whatever ITS
    // level is -- level 0 need not be a filtered one -- and before the k == n return
    if (any_timeout && l == 0 && b == 0 && tid == 0) atomicAdd(status + 1, 1u);
    if (k == n) {
        for (uint32_t i = tid; i < n; i += nt) out[i] = (int32_t)i;
        return;
    }
    const bool in_plan = a.plan.grp[l] != 0;
    const bool timed_out = any_timeout && in_plan;
    const bool filtered = in_plan && !timed_out;
    if (in_plan && a.t.layout == IA_LAYOUT_NHWC) {
        // state of the fused launch (k_rowmax_filter_nhwc) back to zero: the segment's group words
        // -- the straddling ones too -- and its granule; nobody reads them any more in this call
        const int lg = 31 - __builtin_clz((unsigned)a.plan.grp[l]);
        const int64_t gpi = ((int64_t)n + a.plan.grp[l] - 1) >> lg;
        const int64_t g0 = (int64_t)b * gpi, g1 = g0 + gpi;
        uint32_t *gw = const_cast<uint32_t *>(a.groupmax) + a.plan.goff[l];
        for (int64_t j = g0 + tid; j < g1; j += nt) gw[j] = 0u;
        if (tid == 0) a.seg_v[(size_t)b * a.t.num_levels + l] = 0ull;
    }
    const float *src = a.rowmax + (size_t)b * a.anchors_per_img + a.t.anchor_off[l];
    const uint64_t *list = a.cand + (size_t)b * a.anchors_per_img + a.t.anchor_off[l];
    const bool natural = a.t.layout == IA_LAYOUT_NHWC;
    const uint32_t HW = (uint32_t)(a.t.H[l] * a.t.W[l]), A = (uint32_t)a.t.A;
    // candidates: the filter's per-chunk slices, or the whole (small) level
    uint32_t m = n;
    uint32_t nchunks = 0;
    // The head of every chunk slice is requested BEFORE the chunk counts are known (the address
    // does not depend on them): counts -> prefix -> slice loads, a wavefront walking its 2-3 chunks
    // one after the other, was four dependent memory round trips, 4 of the kernel's 6.7 us.
    constexpr int kHead = 4;                       // chunks per wavefront covered by the early loads
    uint64_t head[kHead];
    if (filtered) {
        nchunks = (uint32_t)(a.plan.chunk_off[l + 1] - a.plan.chunk_off[l]);
#pragma unroll
        for (int u = 0; u < kHead; ++u) {
            const uint32_t c = (tid >> 6) + (uint32_t)u * (nt >> 6);
            head[u] = list[(size_t)(c < nchunks ? c : nchunks - 1) * kSelChunk + (tid & 63u)];
        }
        const uint32_t *cc = a.chunk_count + (size_t)b * a.total_chunks + a.plan.chunk_off[l];
        if (tid < (uint32_t)kWave) {               // exclusive prefix of the chunk counts, wave 0
            uint32_t carry = 0;
            for (uint32_t c0 = 0; c0 < nchunks; c0 += kWave) {
                const uint32_t c = c0 + tid;
                const uint32_t v = c < nchunks ? cc[c] : 0u;
                uint32_t incl = v;
                for (int off = 1; off < kWave; off <<= 1) {
                    const uint32_t u = (uint32_t)__shfl_up((int)incl, off);
                    if ((int)tid >= off) incl += u;
                }
                if (c < nchunks) s_pre[c] = carry + incl - v;
                carry += (uint32_t)__shfl((int)incl, kWave - 1);
            }
            if (tid == 0) s_pre[nchunks] = carry;
        }
        __syncthreads();
        m = s_pre[nchunks];
    }
    const uint32_t *pre = s_pre;
    auto cand_key = [=](uint32_t i) -> uint64_t {
        uint32_t lo = 0, hi = nchunks;                     // pre[lo] <= i < pre[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pre[mid] <= i) lo = mid; else hi = mid;
        }
        return list[(size_t)lo * kSelChunk + (i - pre[lo])];
    };
    auto dense_key = [=](uint32_t i) -> uint64_t {
        const uint32_t an = i / HW, p = i - an * HW;
        const uint32_t idx = natural ? i : (p * A + an);
        return ((uint64_t)ordered_key(src[i]) << 32) | (uint64_t)(0xffffffffu - idx);
    };
    bool general = m > (uint32_t)a.lds_cap;
    if (!general) {
        // ---- stage + minimum / maximum of the score words
        uint32_t lo = 0xffffffffu, hi = 0u;
        if (filtered) {                            // a wavefront per chunk slice
            auto put = [&](uint32_t pos, uint64_t x) {
                stage[pos] = x;
                const uint32_t w = (uint32_t)(x >> 32);
                lo = w < lo ? w : lo; hi = w > hi ? w : hi;
            };
            uint32_t c = tid >> 6;
#pragma unroll
            for (int u = 0; u < kHead; ++u, c += nt >> 6) {
                if (c >= nchunks) break;
                const uint32_t base = pre[c], cnt = pre[c + 1] - base;
                const uint32_t j0 = tid & 63u;
                if (j0 < cnt) put(base + j0, head[u]);
                for (uint32_t j = j0 + kWave; j < cnt; j += kWave) put(base + j, list[(size_t)c * kSelChunk + j]);
            }
            for (; c < nchunks; c += nt >> 6) {
                const uint32_t base = pre[c], cnt = pre[c + 1] - base;
                for (uint32_t j = tid & 63u; j < cnt; j += kWave) put(base + j, list[(size_t)c * kSelChunk + j]);
            }
        } else {
            constexpr int DU = 10;                 // unpredicated loads, DU in flight per thread (P5: one round)
            for (uint32_t base = 0; base < m; base += nt * DU) {
                uint64_t xs[DU];
#pragma unroll
                for (int u = 0; u < DU; ++u) {
                    const uint32_t i = base + (uint32_t)u * nt + tid;
                    xs[u] = dense_key(i < m ? i : m - 1);
                }
#pragma unroll
                for (int u = 0; u < DU; ++u) {
                    const uint32_t i = base + (uint32_t)u * nt + tid;
                    if (i < m) {
                        stage[i] = xs[u];
                        const uint32_t w = (uint32_t)(xs[u] >> 32);
                        lo = w < lo ? w : lo; hi = w > hi ? w : hi;
                    }
                }
            }
        }
        for (uint32_t i = tid; i < kBins; i += nt) sc.hist[i] = 0;
        if (tid < 8) sc.misc[tid] = 0;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, off), h2 = (uint32_t)__shfl_xor((int)hi, off);
            lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
        }
        if ((tid & 63) == 0) { s_red[tid >> 6] = lo; s_red[16 + (tid >> 6)] = hi; }
        __syncthreads();
        SEL_PROF(1, l, 1);
        for (uint32_t w = 0; w < nt / kWave; ++w) {
            lo = s_red[w] < lo ? s_red[w] : lo;
            hi = s_red[16 + w] > hi ? s_red[16 + w] : hi;
        }
        // ---- histogram over linear bins between them
        const int shift = bin_shift(hi - lo);
        for (uint32_t i = tid; i < m; i += nt)
            atomicAdd(&sc.hist[((uint32_t)(stage[i] >> 32) - lo) >> shift], 1u);
        __syncthreads();
        SEL_PROF(1, l, 2);
        // ---- s_start[bin] = keys in the bins above it (two bins per thread, top bins first);
        // the largest bucket among the bins that reach into the top k
        {
            const uint32_t top = kBins - 1 - 2 * tid;      // this thread: bins top, top - 1
            const uint32_t c0 = sc.hist[top], c1 = sc.hist[top - 1];
            const uint32_t s2 = c0 + c1;
            uint32_t incl = s2;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t u = (uint32_t)__shfl_up((int)incl, off);
                if ((int)(tid & 63u) >= off) incl += u;
            }
            if ((tid & 63u) == 63u) s_red[tid >> 6] = incl;
            __syncthreads();
            uint32_t base = 0;
            for (uint32_t w = 0; w < (tid >> 6); ++w) base += s_red[w];
            const uint32_t st0 = base + incl - s2, st1 = st0 + c0;
            s_start[top] = st0; s_start[top - 1] = st1;
            sc.hist[top] = 0; sc.hist[top - 1] = 0;        // now the buckets' fill counters
            uint32_t big = (st0 < k ? c0 : 0u);
            big = (st1 < k && c1 > big) ? c1 : big;
            if (big > (uint32_t)kBucketCap) sc.misc[5] = 1;
            if (st0 < k && st0 + c0 >= k) sc.misc[6] = st0 + c0;       // slots in use
            if (st1 < k && st1 + c1 >= k) sc.misc[6] = st1 + c1;
        }
        __syncthreads();
        SEL_PROF(1, l, 3);
        if (sc.misc[5]) {
            general = true;                                 // heavy ties: exact radix select below
        } else {
            // ---- counting sort of the keys of those bins: scatter into the bins' buckets ...
            for (uint32_t i = tid; i < m; i += nt) {
                const uint64_t x = stage[i];
                const uint32_t bin = ((uint32_t)(x >> 32) - lo) >> shift;
                const uint32_t st = s_start[bin];
                if (st < k) sel[st + atomicAdd(&sc.hist[bin], 1u)] = x;
            }
            __syncthreads();
            // ... then every key finds its place among the (few) keys of its bucket
            const uint32_t slots = sc.misc[6];
            for (uint32_t s0 = tid; s0 < slots; s0 += nt) {
                const uint64_t x = sel[s0];
                const uint32_t bin = ((uint32_t)(x >> 32) - lo) >> shift;
                const uint32_t st = s_start[bin], en = bin ? s_start[bin - 1] : m;
                uint32_t rank = 0;
                for (uint32_t q = st; q < en; ++q) rank += (sel[q] > x) ? 1u : 0u;
                if (st + rank < k) out[st + rank] = (int32_t)(0xffffffffu - (uint32_t)x);
            }
            SEL_PROF(1, l, 4);
            return;
        }
    }
    // ---- general path: exact MSB-first radix select + sort (ia_block.hpp)
    __syncthreads();
    if (m <= (uint32_t)a.lds_cap)                           // staged above
        block_topk_desc([stage](uint32_t i) -> uint64_t { return stage[i]; }, m, k, sc, sel);
    else if (filtered)
        block_topk_desc(cand_key, m, k, sc, sel);
    else
        block_topk_desc(dense_key, m, k, sc, sel);
    for (uint32_t i = tid; i < k; i += nt)
        out[i] = (int32_t)(0xffffffffu - (uint32_t)sel[i]);
    SEL_PROF(1, l, 15);
}

struct SelLayout { size_t seg_v, chunk_count, groupmax, cand, total; };

static SelLayout layout(const LevelTable &t, const SelPlan &p, int batch)
{
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    SelLayout w;
    const size_t B = (size_t)batch;
    size_t o = 0;
    w.seg_v = o; o = up(o + B * (size_t)t.num_levels * sizeof(uint64_t));
    // (+2: status words of the fused launch's bounded spin: id of the last call that timed out,
    // number of calls that fell back)
    w.chunk_count = o; o = up(o + (B * (size_t)p.chunk_off[IA_MAX_LEVELS] + 2) * sizeof(uint32_t));
    w.groupmax = o; o = up(o + (size_t)p.goff[IA_MAX_LEVELS] * sizeof(uint32_t));
    w.cand = o;
    if (p.chunk_off[IA_MAX_LEVELS] > 0)
        o = up(o + B * (size_t)t.anchor_off[t.num_levels] * sizeof(uint64_t));
    w.total = o > 0 ? o : 256;
    return w;
}

size_t select_workspace_bytes(const LevelTable &t, int batch)
{
    SelPlan p;
    if (make_sel_plan(t, batch, p)) return 0;
    return layout(t, p, batch).total;
}

size_t select_workspace_status_offset(const LevelTable &t, int batch)
{
    SelPlan p;
    if (make_sel_plan(t, batch, p)) return 0;
    return layout(t, p, batch).chunk_count +
           (size_t)batch * (size_t)p.chunk_off[IA_MAX_LEVELS] * sizeof(uint32_t);
}

float *select_workspace_groupmax(const LevelTable &t, int batch, void *workspace)
{
    // (the words are ordered keys, written by the row-max kernels; float * only in the signature)
    SelPlan p;
    if (make_sel_plan(t, batch, p)) return nullptr;
    return reinterpret_cast<float *>(static_cast<char *>(workspace) + layout(t, p, batch).groupmax);
}

// common part of the launchers: argument block, LDS size of the final kernel
static int prepare_select(const LevelTable &t, const float *rowmax, int batch, int32_t *cand_idx,
                          void *workspace, SelArgs &a, uint32_t &p_max, size_t &dyn)
{
    if (batch < 1 || !rowmax || !cand_idx || !workspace) return IA_E_ARG;
    a.t = t; a.rowmax = rowmax; a.cand_idx = cand_idx; a.batch = batch;
    int rc = make_sel_plan(t, batch, a.plan);
    if (rc) return rc;
    const SelLayout w = layout(t, a.plan, batch);
    char *ws = static_cast<char *>(workspace);
    a.groupmax = reinterpret_cast<uint32_t *>(ws + w.groupmax);
    a.chunk_count = reinterpret_cast<uint32_t *>(ws + w.chunk_count);
    a.seg_v = reinterpret_cast<uint64_t *>(ws + w.seg_v);
    a.cand = reinterpret_cast<uint64_t *>(ws + w.cand);
    a.anchors_per_img = t.anchor_off[t.num_levels];
    a.cands_per_img = t.cand_off[t.num_levels];
    a.total_chunks = a.plan.chunk_off[IA_MAX_LEVELS];
    a.call_id = 0; a.spin_limit = g_spin_limit.load(std::memory_order_relaxed);
    // LDS of the final kernel: sel (next_pow2 of the largest k) + staged candidates
    uint32_t kmax = 1;
    for (int l = 0; l < t.num_levels; ++l) {
        const uint32_t n = (uint32_t)(t.anchor_off[l + 1] - t.anchor_off[l]);
        const uint32_t k = (uint32_t)(t.cand_off[l + 1] - t.cand_off[l]);
        if (k < n && k > kmax) kmax = k;
        if (a.plan.chunk_off[l + 1] - a.plan.chunk_off[l] > kMaxSegChunks) return IA_E_ARG;
    }
    p_max = 1;
    while (p_max < kmax) p_max <<= 1;
    a.lds_cap = kSelDenseMax;
    dyn = ((size_t)p_max + kBucketCap + (size_t)a.lds_cap) * sizeof(uint64_t);
    // up to 32 KiB (sel, k <= 4096) + 96 KiB (stage) + the static scratch: above the 64 KiB a
    // kernel gets without asking.  The attribute belongs to the function ON A DEVICE: set it once
    // per device, thread-safe (a process-wide `static bool` was neither).
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    static std::mutex mu;
    static std::vector<char> done;
    std::lock_guard<std::mutex> lock(mu);
    if ((size_t)dev >= done.size()) done.resize((size_t)dev + 1, 0);
    if (!done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_sel_final),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((IA_MAX_NMS_PRE + kBucketCap + kSelDenseMax) * sizeof(uint64_t)));
        if (e != hipSuccess) return (int)e;
        done[dev] = 1;
    }
    return 0;
}

int launch_select(const LevelTable &t, const float *rowmax, int batch, int32_t *cand_idx,
                  void *workspace, hipStream_t s, bool have_groups)
{
    SelArgs a;
    uint32_t p_max;
    size_t dyn;
    int rc = prepare_select(t, rowmax, batch, cand_idx, workspace, a, p_max, dyn);
    if (rc) return rc;
    const int chunks = a.total_chunks;
    if (chunks > 0) {
        if (!have_groups)
            hipLaunchKernelGGL(k_sel_groupmax, dim3((unsigned)((a.plan.goff[IA_MAX_LEVELS] + 255) / 256)),
                               dim3(256), 0, s, a, const_cast<uint32_t *>(a.groupmax));
        hipLaunchKernelGGL(k_sel_filter, dim3((unsigned)chunks, (unsigned)batch),
                           dim3(kFilterThreads), 0, s, a);
        rc = hip_status(hipGetLastError());
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_sel_final, dim3((unsigned)t.num_levels, (unsigned)batch),
                       dim3(kFinalThreads), dyn, s, a, p_max);
    return hip_status(hipGetLastError());
}

int launch_groupmax(const LevelTable &t, const float *rowmax, int batch, void *workspace, hipStream_t s)
{
    SelArgs a;
    uint32_t p_max;
    size_t dyn;
    int32_t dummy = 0;
    int rc = prepare_select(t, rowmax, batch, &dummy, workspace, a, p_max, dyn);     // cand_idx is not touched
    if (rc) return rc;
    if (a.total_chunks > 0) {
        hipLaunchKernelGGL(k_sel_groupmax, dim3((unsigned)((a.plan.goff[IA_MAX_LEVELS] + 255) / 256)),
                           dim3(256), 0, s, a, const_cast<uint32_t *>(a.groupmax));
        rc = hip_status(hipGetLastError());
    }
    return rc;
}

// ia_get_bboxes / ia_decode_stage: row-max + top-k.  Channels-last heads with filtered levels take
// the fused launch (k_rowmax_filter_nhwc) + k_sel_final; everything else the separate kernels.
// `workspace` (select workspace) must have been zeroed once by its owner (seg_done).
int launch_rowmax_select(const LevelTable &t, const ia_level_ptrs &p, int batch, int dtype,
                         float *rowmax, int32_t *cand_idx, void *workspace, hipStream_t s)
{
    SelArgs a;
    uint32_t p_max;
    size_t dyn;
    int rc = prepare_select(t, rowmax, batch, cand_idx, workspace, a, p_max, dyn);
    if (rc) return rc;
    if (dtype != IA_F32 && dtype != IA_BF16) return IA_E_ARG;
    const int ppl = (dtype == IA_F32) ? Lane<float>::PPL : Lane<uint16_t>::PPL;
    // IA_FUSED_ROWMAX_FILTER=0 in the environment: the separate kernels (A/B runs, bisecting)
    static const bool allow_fused = [] {
        const char *e = getenv("IA_FUSED_ROWMAX_FILTER");
        return !(e && e[0] == '0');
    }();
    if (t.softmax) {            // softmax row scores (decode.hip), then the separate selection kernels
        rc = launch_rowmax(t, p, batch, dtype, rowmax, s);
        if (rc) return rc;
        return launch_select(t, rowmax, batch, cand_idx, workspace, s, false);
    }
    bool fused = allow_fused && t.layout == IA_LAYOUT_NHWC && a.total_chunks > 0 && t.C % ppl == 0 &&
                 t.C / ppl <= kMaxVpr;
    for (int l = 0; l < t.num_levels && fused; ++l) fused = (((uintptr_t)p.cls[l] & 15u) == 0);
    RowmaxNhwcArgs ra;
    FusedOrder fo;
    int64_t blocks = 0;
    if (fused) {
        ra.t = t; ra.p = p; ra.rowmax = rowmax; ra.batch = batch;
        ra.anchors_per_img = t.anchor_off[t.num_levels];
        ra.plan = a.plan; ra.groupmax = const_cast<uint32_t *>(a.groupmax);
        ra.big_first = 1;
        for (int i = 0; i <= IA_MAX_LEVELS; ++i) ra.blk_off[i] = 0;        // unused by the fused kernel
        // Launch order: the levels' row-max workgroups, largest level first; the LEADER of a
        // filtered segment (its chunk 0: waits for the segment's group words, derives the threshold,
        // publishes it) right behind the row-max workgroups of the NEXT level, so that the
        // threshold is out before the stream ends -- a handful of waiting workgroups; the other
        // filter workgroups last.  (ALL filter workgroups behind their level: they are dispatched
        // 10-30 us before their data is complete -- dispatch across the XCDs is far from in order
        // -- and the wavefront slots of hundreds of waiting workgroups cost the stream more than
        // the overlap wins: 95 vs 92 us.  All of them last, leaders included: the threshold
        // computation sits in the tail, 91 us under the profiler.)
        fo.item_off[0] = 0;
        int it = 0;
        auto push = [&](int l, int kind, int64_t nb) -> int {
            blocks += nb;
            if (blocks > 2147483647LL) return IA_E_ARG;
            fo.item_level[it] = l; fo.item_filter[it] = kind;
            fo.item_off[++it] = (int32_t)blocks;
            return 0;
        };
        auto chunks = [&](int l) { return (int64_t)(a.plan.chunk_off[l + 1] - a.plan.chunk_off[l]); };
        for (int l = 0; l < IA_MAX_LEVELS; ++l) fo.units[l] = 0;
        for (int l = 0; l < t.num_levels; ++l) {
            const int64_t units = (int64_t)batch * sel_units_per_image(t, l);
            if (units > 2147483647LL) return IA_E_ARG;
            fo.units[l] = (int32_t)units;
            if ((rc = push(l, 0, (units + 3) / 4))) return rc;
            if (l > 0 && chunks(l - 1) > 0 && (rc = push(l - 1, 1, batch))) return rc;
        }
        const int last = t.num_levels - 1;
        if (chunks(last) > 0 && (rc = push(last, 1, batch))) return rc;
        for (int l = 0; l < t.num_levels; ++l)
            if (chunks(l) > 1 && (rc = push(l, 2, (chunks(l) - 1) * batch))) return rc;
        fo.n_items = it;
        for (; it < 3 * IA_MAX_LEVELS; ) { fo.item_level[it] = 0; fo.item_filter[it] = 0; fo.item_off[++it] = (int32_t)blocks; }
    }
    if (!fused) {
        rc = launch_rowmax(t, p, batch, dtype, rowmax, s,
                           reinterpret_cast<float *>(const_cast<uint32_t *>(a.groupmax)));
        if (rc) return rc;
        return launch_select(t, rowmax, batch, cand_idx, workspace, s, true);
    }
    const dim3 grid((unsigned)blocks), block(kFilterThreads);
    const int vpr = t.C / ppl;
    uint32_t id = ++g_fused_calls;
    if (id == 0u) id = ++g_fused_calls;                    // 0 = "not a fused launch"
    a.call_id = id;
    if (dtype == IA_F32) {
        if (vpr == 20) hipLaunchKernelGGL((k_rowmax_filter_nhwc<float, 20>), grid, block, 0, s, ra, a, fo);
        else hipLaunchKernelGGL((k_rowmax_filter_nhwc<float, 0>), grid, block, 0, s, ra, a, fo);
    } else {
        if (vpr == 10) hipLaunchKernelGGL((k_rowmax_filter_nhwc<uint16_t, 10>), grid, block, 0, s, ra, a, fo);
        else hipLaunchKernelGGL((k_rowmax_filter_nhwc<uint16_t, 0>), grid, block, 0, s, ra, a, fo);
    }
    rc = hip_status(hipGetLastError());
    if (!rc) {
        hipLaunchKernelGGL(k_sel_final, dim3((unsigned)t.num_levels, (unsigned)batch),
                           dim3(kFinalThreads), dyn, s, a, p_max);
        rc = hip_status(hipGetLastError());
    }
    if (rc) {
        // k_sel_final did not run: the group words / granules of the fused launch may be set and
        // would let the next call skip its waits -- back to the zero state of the contract
        const SelPlan &pl = a.plan;
        (void)hipMemsetAsync(workspace, 0, layout(t, pl, batch).cand, s);
    }
    return rc;
}

}  // namespace ia

/* tests: bound of the fused launch's spins (negative: the default).  0 makes every wait give up at
 * once, i.e. forces the fallback of k_sel_final. */
extern "C" int ia_debug_fused_spin_limit(int64_t limit)
{
    ia::g_spin_limit.store(limit < 0 ? ia::kSpinLimit : (limit > 0xffffffffLL ? 0xffffffffu : (uint32_t)limit),
                           std::memory_order_relaxed);
    return 0;
}
