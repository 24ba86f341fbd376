// Workgroup-level primitives for gfx950 (wave64): radix top-k selection over a
// virtual array of unique 64-bit keys, LDS bitonic sort, wave-aggregated LDS
// histogram updates.  Used by the per-level top-k (select.hip) and the
// per-image final top-k (nms.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef IA_BLOCK_PROF
#define IA_BLOCK_PROF(i) do { } while (0)      // phase timestamps of tools/ubench/select_bench.hip
#endif

namespace ia {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t lane_prefix_popc(uint64_t mask)
{
    uint32_t lo = (uint32_t)mask, hi = (uint32_t)(mask >> 32);
    return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
}

// LDS histogram increment with wave-level aggregation of the most common
// digits: degenerate inputs (all keys equal -> one hot bin) would otherwise
// serialise 64-way on a single LDS address.
__device__ __forceinline__ void hist_add(uint32_t *hist, bool active, uint32_t digit)
{
    uint64_t act = __ballot(active);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (act == 0) break;
        int leader = __builtin_ctzll(act);
        uint32_t d0 = (uint32_t)__shfl((int)digit, leader);
        uint64_t same = __ballot(active && digit == d0);
        if (lane_id() == leader) atomicAdd(&hist[d0], (uint32_t)__builtin_popcountll(same));
        if (active && digit == d0) active = false;
        act &= ~same;
    }
    if (active) atomicAdd(&hist[digit], 1u);
}

// Descending bitonic sort of P (power of two) 64-bit keys in LDS by the whole
// workgroup.  Caller must __syncthreads() before (keys written) -- the routine
// ends with a barrier.
__device__ __forceinline__ void bitonic_sort_desc(uint64_t *keys, uint32_t P)
{
    const uint32_t nt = blockDim.x * blockDim.y;
    const uint32_t tid = threadIdx.y * blockDim.x + threadIdx.x;
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t t = tid; t < (P >> 1); t += nt) {
                uint32_t pos = 2 * t - (t & (stride - 1));
                uint32_t par = pos + stride;
                bool up = ((pos & size) == 0);
                uint64_t a = keys[pos], b = keys[par];
                if ((a < b) == up) { keys[pos] = b; keys[par] = a; }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t v)
{
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct TopkScratch {
    uint32_t hist[2048];
    uint32_t misc[8];   // 0: digit, 1: count above, 2: count in digit, 3: collect counter, 4: tie cut
};

constexpr uint32_t kTieShortcut = 512;   // ties resolved by rank counting instead of radix passes

// Select the k largest of n UNIQUE 64-bit keys key(i), i in [0,n), and leave
// them sorted descending in sel[0..k).  sel must hold next_pow2(k) entries.
// 1 <= k <= n.  Whole workgroup participates (blockDim.x*blockDim.y threads,
// a multiple of 64, >= 64).  MSB-first radix select with 11/11/10-bit digits;
// stops as soon as the digit bin equals the remaining need, so tie-free score
// keys never touch the low (index) half.  When the high words (scores) tie across the cut
// and at most kTieShortcut keys share the threshold score, the cut inside the tie group is
// found by rank counting in LDS (one compaction pass + n_tie broadcast reads per thread)
// instead of three more radix passes over all n keys: random-init networks put a handful of
// equal scores on every cut, which used to double the kernel's time.
template <class KeyFn>
__device__ void block_topk_desc(KeyFn key, uint32_t n, uint32_t k, TopkScratch &sc, uint64_t *sel)
{
    const uint32_t nt = blockDim.x * blockDim.y;
    const uint32_t tid = threadIdx.y * blockDim.x + threadIdx.x;
    uint64_t prefix = 0;
    int bits_done = 0;
    uint32_t need = k;
    uint64_t cut = 0;                 // final answer: the selected keys are exactly those >= cut
    bool have_cut = false;
    const int widths[6] = {11, 11, 10, 11, 11, 10};
    for (int pass = 0; pass < 6; ++pass) {
        const int wb = widths[pass];
        const uint32_t nbins = 1u << wb;
        const int shift = 64 - bits_done - wb;
        for (uint32_t i = tid; i < nbins; i += nt) sc.hist[i] = 0;
        __syncthreads();
        // U keys per thread are fetched before any of them is binned: the key functor may be a
        // global load, and one load per ballot round would leave the pass latency-bound
        constexpr int U = 8;
        for (uint32_t base = 0; base < n; base += nt * U) {
            uint64_t xs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = base + u * nt + tid;
                xs[u] = key(i < n ? i : n - 1);     // never predicated: see select.hip
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = base + u * nt + tid;
                if (base + u * nt >= n) break;             // uniform
                bool act = i < n;
                const uint64_t x = xs[u];
                if (bits_done > 0) act = act && ((x >> (64 - bits_done)) == prefix);
                hist_add(sc.hist, act, (uint32_t)(x >> shift) & (nbins - 1));
            }
        }
        __syncthreads();
        // find the digit d (from the top) where the running count reaches `need`
        if (tid < kWave) {
            const uint32_t per = nbins / kWave;
            const uint32_t hi_bin = nbins - per * tid;      // exclusive upper bin of this lane
            uint32_t s = 0;
            for (uint32_t j = 0; j < per; ++j) s += sc.hist[hi_bin - 1 - j];
            uint32_t incl = s;
            for (int off = 1; off < kWave; off <<= 1) {
                uint32_t v = (uint32_t)__shfl_up((int)incl, off);
                if ((int)tid >= off) incl += v;
            }
            uint64_t reach = __ballot(incl >= need);
            int first = __builtin_ctzll(reach);            // reach != 0 because total >= need
            if ((int)tid == first) {
                uint32_t above = incl - s;
                uint32_t d = hi_bin - 1;
                for (uint32_t j = 0; j < per; ++j) {
                    uint32_t c = sc.hist[hi_bin - 1 - j];
                    if (above + c >= need) { d = hi_bin - 1 - j; sc.misc[2] = c; break; }
                    above += c;
                }
                sc.misc[0] = d;
                sc.misc[1] = above;
            }
        }
        __syncthreads();
        const uint32_t d = sc.misc[0], above = sc.misc[1], in_d = sc.misc[2];
        need -= above;
        prefix = (prefix << wb) | d;
        bits_done += wb;
        __syncthreads();
        IA_BLOCK_PROF(2 + pass);
        if (in_d == need) break;
        if (bits_done == 32 && in_d <= kTieShortcut) {
            // in_d keys share the threshold's high word, `need` of them (the largest low words)
            // are wanted: compact their low words into the (now free) histogram array and let
            // every tie count how many ties lie above it
            if (tid == 0) sc.misc[3] = 0;
            __syncthreads();
            for (uint32_t base = 0; base < n; base += nt * U) {
                uint64_t xs[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t i = base + u * nt + tid;
                    xs[u] = key(i < n ? i : n - 1);     // never predicated: see select.hip
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t i = base + u * nt + tid;
                    if (base + u * nt >= n) break;         // uniform
                    const bool tie = (i < n) && ((xs[u] >> 32) == prefix);
                    const uint64_t mt = __ballot(tie);
                    if (mt) {
                        uint32_t b0 = 0;
                        const int leader = __builtin_ctzll(mt);
                        if (lane_id() == leader)
                            b0 = atomicAdd(&sc.misc[3], (uint32_t)__builtin_popcountll(mt));
                        b0 = (uint32_t)__shfl((int)b0, leader);
                        if (tie) sc.hist[b0 + lane_prefix_popc(mt)] = (uint32_t)xs[u];
                    }
                }
            }
            __syncthreads();
            for (uint32_t j = tid; j < in_d; j += nt) {
                const uint32_t lo = sc.hist[j];
                uint32_t rank = 0;
                for (uint32_t q = 0; q < in_d; ++q) rank += (sc.hist[q] > lo) ? 1u : 0u;
                if (rank == need - 1) sc.misc[4] = lo;     // unique keys: exactly one writer
            }
            __syncthreads();
            cut = (prefix << 32) | (uint64_t)sc.misc[4];
            have_cut = true;
            __syncthreads();
            IA_BLOCK_PROF(8);
            break;
        }
    }
    if (!have_cut) cut = (bits_done >= 64) ? prefix : (prefix << (64 - bits_done));
    // collect everything >= cut: exactly k keys
    if (tid == 0) sc.misc[3] = 0;
    const uint32_t P = next_pow2(k);
    for (uint32_t i = tid; i < P; i += nt) sel[i] = 0;
    __syncthreads();
    constexpr int U2 = 8;
    for (uint32_t base = 0; base < n; base += nt * U2) {
        uint64_t xs[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const uint32_t i = base + u * nt + tid;
            xs[u] = key(i < n ? i : n - 1);     // never predicated: see select.hip
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const uint32_t i = base + u * nt + tid;
            if (base + u * nt >= n) break;                 // uniform
            const uint64_t x = xs[u];
            bool take = (i < n) && (x >= cut);
            uint64_t m = __ballot(take);
            if (m) {
                uint32_t b0 = 0;
                int leader = __builtin_ctzll(m);
                if (lane_id() == leader) b0 = atomicAdd(&sc.misc[3], (uint32_t)__builtin_popcountll(m));
                b0 = (uint32_t)__shfl((int)b0, leader);
                if (take) {
                    uint32_t pos = b0 + lane_prefix_popc(m);
                    if (pos < P) sel[pos] = x;
                }
            }
        }
    }
    __syncthreads();
    IA_BLOCK_PROF(9);
    bitonic_sort_desc(sel, P);
    IA_BLOCK_PROF(10);
}

}  // namespace ia
