// Batched greedy NMS for all (image, class) problems, and the per-image final
// top-k.  Replaces multiclass_nms (reference
// mmdet/core/post_processing/bbox_nms.py:33-56) and the native op
// nms_cpu_kernel (reference mmdet/ops/nms/src/nms_cpu.cpp:4-59; ">=" at :55,
// ascending kept indices at :58).  The reference CUDA path (nms_kernel.cu)
// copies a bitmask to the host for every call; here everything stays on the
// device.
//
// Key observation: the boxes are class-agnostic, so the relation
// "IoU(i, j) >= thr" is the SAME for all 80 class problems of an image; only
// the subset (score > score_thr) and the order (by class score) differ.
//
//   k_adj          once per image: the symmetric R x R suppression relation as
//                  a bit matrix (64x64 tiles, one wavefront each, v_readlane
//                  broadcasts of the column boxes).  Rows / columns whose best
//                  class score is <= score_thr never take part and are skipped.
//   k_nms_resolve  one small workgroup per (image, class): wave64-ballot
//                  compaction of the class column, LDS bitonic sort ->
//                  canonical order (score desc, row asc); then chunks of 64
//                  sorted candidates: a candidate is suppressed iff its
//                  adjacency row ANDed with the bitmap of already kept rows is
//                  non-zero; inside a chunk the 64x64 sub-relation is extracted
//                  from the same words and resolved with ballot masks.  No box
//                  arithmetic at all: ~n*W word loads per class instead of
//                  n*kept pair tests (random-init case: 3.4 ms -> see DESIGN.md).
//   Blocks of image b are placed on XCD b % 8 (block id = c * Bpad + b), so an
//   image's 3 MB bit matrix stays in ONE 4 MiB L2.
//
// The suppression test must equal the reference's fp32
// `inter / (iarea + areas[j] - inter) >= thr` decision bit for bit.  A
// correctly rounded fp32 quotient q satisfies fl(q) >= thr  <=>  q >= mid
// (or > mid), where mid is the midpoint between thr and its fp32 predecessor;
// for union > 0 that is `inter >= mid * union`, exact in fp64 (25 + 24
// significant bits), so no division is needed on the hot path.
#include <math.h>
#include <string.h>
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_block.hpp"
#include "ia_nms.hpp"

namespace ia {

constexpr int kResThreads = 256;              // k_nms_resolve workgroup
constexpr int kResWaves = kResThreads / kWave;
constexpr int kAdjGroup = 16;                 // adjacency row words are padded to a multiple of this

// value of lane `src` (wave-uniform index) in every lane: v_readlane_b32, no LDS round trip
__device__ __forceinline__ float bcast(float v, int src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}


// ------------------------------------------------------------------ adjacency bit matrix
// adj[(b*R + r) * W + w] bit c  <=>  box r and box 64*w + c suppress each other.
struct AdjArgs {
    const float *boxes;          // (B, R, stride) fp32, x1 y1 x2 y2 first
    const float *best_score;     // (B, R) best class score per row, or NULL (all rows active)
    uint64_t *adj;
    IouThr thr;
    float score_thr;
    int32_t R, W, stride;
    const int32_t *gate;         // optional (B): images with gate[b] == 0 are skipped
};

__global__ void __launch_bounds__(64) k_adj(AdjArgs a)
{
    const int lane = threadIdx.x;
    const int tj = blockIdx.x, ti = blockIdx.y, b = blockIdx.z;
    if (a.gate && !a.gate[b]) return;
    const int r = ti * 64 + lane, c = tj * 64 + lane;
    const float *bx = a.boxes + (size_t)b * a.R * a.stride;
    const float *bs = a.best_score ? a.best_score + (size_t)b * a.R : nullptr;
    const bool row_ok = r < a.R && (!bs || bs[r] > a.score_thr);
    const bool col_ok = c < a.R && (!bs || bs[c] > a.score_thr);
    const uint64_t colmask = __ballot(col_ok);
    if (colmask == 0 || __ballot(row_ok) == 0) return;           // wave-uniform
    float rx1 = 0, ry1 = 0, rx2 = 0, ry2 = 0, cx1 = 0, cy1 = 0, cx2 = 0, cy2 = 0;
    if (r < a.R) { const float *q = bx + (size_t)r * a.stride; rx1 = q[0]; ry1 = q[1]; rx2 = q[2]; ry2 = q[3]; }
    if (c < a.R) { const float *q = bx + (size_t)c * a.stride; cx1 = q[0]; cy1 = q[1]; cx2 = q[2]; cy2 = q[3]; }
    const float rar = ((rx2 - rx1) + 1.0f) * ((ry2 - ry1) + 1.0f);   // nms_cpu.cpp:18
    const float car = ((cx2 - cx1) + 1.0f) * ((cy2 - cy1) + 1.0f);
    uint64_t word = 0;
    uint64_t todo = colmask;
    while (todo) {
        const int cc = __builtin_ctzll(todo);
        todo &= todo - 1;
        const float sx1 = bcast(cx1, cc), sy1 = bcast(cy1, cc), sx2 = bcast(cx2, cc);
        const float sy2 = bcast(cy2, cc), sar = bcast(car, cc);
        if (suppresses(sx1, sy1, sx2, sy2, sar, rx1, ry1, rx2, ry2, rar, a.thr))
            word |= 1ull << cc;
    }
    if (row_ok) a.adj[((size_t)b * a.R + r) * a.W + tj] = word;
}

static int launch_adj(const float *boxes, int stride, const float *best_score, int batch, int R,
                      int W, const IouThr &thr, float score_thr, uint64_t *adj, hipStream_t s,
                      const int32_t *gate = nullptr)
{
    AdjArgs a;
    a.gate = gate;
    a.boxes = boxes; a.best_score = best_score; a.adj = adj; a.thr = thr; a.score_thr = score_thr;
    a.R = R; a.W = W; a.stride = stride;
    const unsigned tiles = (unsigned)((R + 63) / 64);
    hipLaunchKernelGGL(k_adj, dim3(tiles, tiles, (unsigned)batch), dim3(64), 0, s, a);
    return hip_status(hipGetLastError());
}

// ------------------------------------------------------------------ per-problem order
// k_class_sort: rows with score > score_thr of one (image, class), in canonical order
// (score descending, row ascending) -> sorted_rows (uint16), n_in.
constexpr int kSortThreads = 1024;

struct ClassSrc {                       // class-major fused scores of one problem
    const float *sc;
    float thr;
    __device__ __forceinline__ bool pred(uint32_t r) const { return sc[r] > thr; }   // bbox_nms.py:34
    __device__ __forceinline__ float score(uint32_t r) const { return sc[r]; }
};
struct DetsSrc {                        // (n,5) dets of the standalone op: every row takes part
    const float *d;
    __device__ __forceinline__ bool pred(uint32_t) const { return true; }
    __device__ __forceinline__ float score(uint32_t r) const { return d[5 * (size_t)r + 4]; }
};

template <class Src>
__device__ void class_sort_block(uint32_t R, const Src &src, uint16_t *rows_out, int32_t *n_out,
                                 uint64_t *keys)
{
    __shared__ uint32_t counter;
    const uint32_t tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    if (tid == 0) counter = 0;
    __syncthreads();
    const uint32_t R_up = (R + kWave - 1) & ~(uint32_t)(kWave - 1);
    for (uint32_t r = tid; r < R_up; r += kSortThreads) {
        bool in = (r < R) && src.pred(r);
        uint64_t m = __ballot(in);
        if (m) {
            uint32_t base = 0;
            int leader = __builtin_ctzll(m);
            if (lane == leader) base = atomicAdd(&counter, (uint32_t)__builtin_popcountll(m));
            base = (uint32_t)__shfl((int)base, leader);
            if (in)
                keys[base + lane_prefix_popc(m)] =
                    ((uint64_t)ordered_key(src.score(r)) << 32) | (uint64_t)(0xffffffffu - r);
        }
    }
    __syncthreads();
    const uint32_t n = counter;
    if (tid == 0) *n_out = (int32_t)n;
    if (n == 0) return;
    const uint32_t P = next_pow2(n < 2 ? 2 : n);
    for (uint32_t i = n + tid; i < P; i += kSortThreads) keys[i] = 0;
    __syncthreads();
    bitonic_sort_desc(keys, P);
    for (uint32_t i = tid; i < n; i += kSortThreads)
        rows_out[i] = (uint16_t)(0xffffffffu - (uint32_t)keys[i]);
}

struct SortArgs {
    const float *scores_t;
    uint16_t *sorted_rows;
    int32_t *n_in;
    float score_thr;
    int32_t R, Rs, C;
    const int32_t *gate;
};

__global__ void __launch_bounds__(kSortThreads) k_class_sort(SortArgs a)
{
    extern __shared__ uint64_t keys[];
    const int c = blockIdx.x, b = blockIdx.y;
    if (a.gate && !a.gate[b]) return;
    const size_t prob = (size_t)b * a.C + c;
    ClassSrc src{a.scores_t + prob * a.Rs, a.score_thr};
    class_sort_block((uint32_t)a.R, src, a.sorted_rows + prob * a.Rs, a.n_in + prob, keys);
}

struct SortSingleArgs {
    const float *dets;
    uint16_t *sorted_rows;
    int32_t *n_in;
    int32_t n;
};

__global__ void __launch_bounds__(kSortThreads) k_dets_sort(SortSingleArgs a)
{
    extern __shared__ uint64_t keys[];
    DetsSrc src{a.dets};
    class_sort_block((uint32_t)a.n, src, a.sorted_rows, a.n_in, keys);
}

// ------------------------------------------------------------------ per-problem resolve
// 4 wavefronts per problem.  For each chunk of 64 sorted candidates the adjacency row words
// are split between the wavefronts (independent 16-byte loads in flight on all four), each
// produces a partial "suppressed by an already kept row" flag and partial in-chunk neighbour
// mask per candidate; wavefront 0 ORs them and runs the (pure bit-mask) greedy step.
constexpr int kWordGroup = 8;           // adjacency words handled per load batch (4 x 16 B per lane)

struct ResSmem {
    uint64_t kept[IA_MAX_CANDIDATES / 64];     // bitmap of kept ROWS (also the output set)
    uint64_t pE[kResWaves][kWave];
    uint32_t psup[kResWaves][kWave];
    uint32_t wave_cnt[kResWaves];
};

__device__ uint32_t nms_resolve_block(uint32_t R, uint32_t W, uint32_t n, const uint16_t *rows,
                                      const uint64_t *adj, int32_t *keep_out, ResSmem &sm)
{
    const uint32_t tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    for (uint32_t i = tid; i < IA_MAX_CANDIDATES / 64; i += kResThreads) sm.kept[i] = 0;
    __syncthreads();
    if (n == 0) return 0;
    const uint32_t nchunks = (n + kWave - 1) / kWave;
    const uint32_t ngroups = W / kWordGroup;
    for (uint32_t k = 0; k < nchunks; ++k) {
        const uint32_t spos = k * kWave + lane;
        const bool valid = spos < n;
        const uint32_t r = valid ? rows[spos] : 0u;
        const uint32_t ws = r >> 6, bsft = r & 63u;
        const uint64_t *rowp = adj + (size_t)r * W;
        bool sup = false;
        uint64_t E = 0;                                    // earlier-in-chunk neighbours of this lane
        for (uint32_t gi = wave; gi < ngroups; gi += kResWaves) {
            const uint32_t g = gi * kWordGroup;
            const uint64_t kw_l = (lane < kWordGroup) ? sm.kept[g + lane] : 0ull;
            const uint64_t has_kept = __ballot(kw_l != 0);
            const uint64_t mgrp = __ballot(valid && (ws / kWordGroup) == gi);
            if (has_kept == 0 && mgrp == 0) continue;      // uniform: nothing to test in this group
            uint64_t v[kWordGroup];
#pragma unroll
            for (int i = 0; i < kWordGroup / 2; ++i) {
                ulonglong2 q = valid ? reinterpret_cast<const ulonglong2 *>(rowp + g)[i]
                                     : make_ulonglong2(0ull, 0ull);
                v[2 * i] = q.x; v[2 * i + 1] = q.y;
            }
#pragma unroll
            for (int i = 0; i < kWordGroup; ++i) {
                const uint64_t kw = sm.kept[g + i];         // uniform LDS broadcast
                if (v[i] & kw) sup = true;
                uint64_t mw = __ballot(valid && ws == g + i);   // chunk members living in this word
                while (mw) {
                    const int s2 = __builtin_ctzll(mw);
                    mw &= mw - 1;
                    const uint32_t b2 = (uint32_t)__builtin_amdgcn_readlane((int)bsft, s2);
                    E |= ((v[i] >> b2) & 1ull) << s2;
                }
            }
        }
        sm.psup[wave][lane] = sup ? 1u : 0u;
        sm.pE[wave][lane] = E;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 1; w < kResWaves; ++w) {
                sup = sup || (sm.psup[w][lane] != 0);
                E |= sm.pE[w][lane];
            }
            E &= (1ull << lane) - 1ull;                    // sources earlier in the order only
            uint64_t amask = __ballot(valid && !sup);
            uint64_t todo = amask;
            while (todo) {
                const int s2 = __builtin_ctzll(todo);
                todo &= todo - 1;
                if (!((amask >> s2) & 1ull)) continue;
                amask &= ~__ballot(((E >> s2) & 1ull) != 0);
            }
            if ((amask >> lane) & 1ull)
                atomicOr(reinterpret_cast<unsigned long long *>(&sm.kept[ws]), 1ull << bsft);
        }
        __syncthreads();
    }

    // emit kept rows ascending (nms_cpu.cpp:58)
    const uint32_t R_up = (R + kWave - 1) & ~(uint32_t)(kWave - 1);
    uint32_t total = 0;
    for (uint32_t r0 = 0; r0 < R_up; r0 += kResThreads) {
        const uint32_t r = r0 + tid;
        const bool kp = (r < R) && ((sm.kept[r >> 6] >> (r & 63u)) & 1ull);
        const uint64_t m = __ballot(kp);
        if (lane == 0) sm.wave_cnt[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kResWaves; ++w) {
            const uint32_t cnt = sm.wave_cnt[w];
            before += (w < wave) ? cnt : 0;
            all += cnt;
        }
        if (kp) keep_out[total + before + lane_prefix_popc(m)] = (int32_t)r;
        total += all;
        __syncthreads();
    }
    return total;
}

struct NmsArgs {
    const uint16_t *sorted_rows;
    const int32_t *n_in;
    const uint64_t *adj;
    int32_t *keep_count;
    int32_t *keep_rows;
    int32_t R, Rs, C, W, B, Bpad;
    const int32_t *gate;
};

__global__ void __launch_bounds__(kResThreads) k_nms_resolve(NmsArgs a)
{
    __shared__ ResSmem sm;
    const int c = blockIdx.x / a.Bpad, b = blockIdx.x - c * a.Bpad;   // XCD = blockIdx % 8 = b % 8
    if (b >= a.B || (a.gate && !a.gate[b])) return;
    const size_t prob = (size_t)b * a.C + c;
    uint32_t cnt = nms_resolve_block((uint32_t)a.R, (uint32_t)a.W, (uint32_t)a.n_in[prob],
                                     a.sorted_rows + prob * a.Rs, a.adj + (size_t)b * a.R * a.W,
                                     a.keep_rows + prob * a.Rs, sm);
    if (threadIdx.x == 0) a.keep_count[prob] = (int32_t)cnt;
}

static uint32_t pow2_at_least(uint32_t v)
{
    uint32_t p = 2;
    while (p < v) p <<= 1;
    return p;
}

int nms_adj_words(int R) { return ((R + 63) / 64 + kAdjGroup - 1) / kAdjGroup * kAdjGroup; }

// workspace of the NMS stage: adjacency bit matrix | sorted rows (uint16) | participant counts
size_t nms_workspace_bytes(int batch, int R, int C, size_t off[3])
{
    const size_t Rs = (size_t)(R + 63) / 64 * 64;
    size_t o = 0;
    off[0] = o; o += ((size_t)batch * R * nms_adj_words(R) * sizeof(uint64_t) + 255) / 256 * 256;
    off[1] = o; o += ((size_t)batch * C * Rs * sizeof(uint16_t) + 255) / 256 * 256;
    off[2] = o; o += ((size_t)batch * C * sizeof(int32_t) + 255) / 256 * 256;
    return o;
}

int launch_nms(const float *boxes, const float *scores_t, const float *best_score, int batch,
               int R, int Rs, int C, float score_thr, float iou_thr, void *workspace,
               int32_t *keep_count, int32_t *keep_rows, hipStream_t s, const int32_t *gate)
{
    if (R > IA_MAX_CANDIDATES) return IA_E_LIMIT_BOXES;
    if (batch < 1 || R < 1 || C < 1 || Rs < R) return IA_E_ARG;
    if (!boxes || !scores_t || !workspace || !keep_count || !keep_rows) return IA_E_ARG;
    size_t off[3];
    nms_workspace_bytes(batch, R, C, off);
    char *ws = static_cast<char *>(workspace);
    uint64_t *adj = reinterpret_cast<uint64_t *>(ws + off[0]);
    uint16_t *sorted_rows = reinterpret_cast<uint16_t *>(ws + off[1]);
    int32_t *n_in = reinterpret_cast<int32_t *>(ws + off[2]);
    const int W = nms_adj_words(R);
    int rc = launch_adj(boxes, 4, best_score, batch, R, W, make_thr(iou_thr), score_thr, adj, s, gate);
    if (rc) return rc;
    SortArgs sa;
    sa.gate = gate;
    sa.scores_t = scores_t; sa.sorted_rows = sorted_rows; sa.n_in = n_in; sa.score_thr = score_thr;
    sa.R = R; sa.Rs = Rs; sa.C = C;
    size_t lds = sizeof(uint64_t) * (size_t)pow2_at_least((uint32_t)R);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_class_sort),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_class_sort, dim3((unsigned)C, (unsigned)batch), dim3(kSortThreads), lds, s, sa);
    if ((rc = hip_status(hipGetLastError()))) return rc;
    NmsArgs a;
    a.gate = gate;
    a.sorted_rows = sorted_rows; a.n_in = n_in; a.adj = adj; a.keep_count = keep_count;
    a.keep_rows = keep_rows; a.R = R; a.Rs = Rs; a.C = C; a.W = W; a.B = batch;
    a.Bpad = (batch + 7) / 8 * 8;
    hipLaunchKernelGGL(k_nms_resolve, dim3((unsigned)(C * a.Bpad)), dim3(kResThreads), 0, s, a);
    return hip_status(hipGetLastError());
}

// ------------------------------------------------------------------ standalone op
struct NmsSingleArgs {
    const uint16_t *sorted_rows;
    const int32_t *n_in;
    const uint64_t *adj;
    int32_t *keep;
    int32_t *count;
    int32_t n, W;
};

__global__ void __launch_bounds__(kResThreads) k_nms_single(NmsSingleArgs a)
{
    __shared__ ResSmem sm;
    uint32_t cnt = nms_resolve_block((uint32_t)a.n, (uint32_t)a.W, (uint32_t)a.n_in[0],
                                     a.sorted_rows, a.adj, a.keep, sm);
    if (threadIdx.x == 0) *a.count = (int32_t)cnt;
}

size_t nms_single_workspace_bytes(int n)
{
    if (n < 1) return 0;
    size_t off[3];
    return nms_workspace_bytes(1, n, 1, off);
}

int launch_nms_single(const float *dets, int n, float iou_thr, int32_t *keep, int32_t *count,
                      void *workspace, size_t workspace_bytes, hipStream_t s)
{
    if (n > IA_MAX_CANDIDATES) return IA_E_LIMIT_BOXES;
    if (n < 0 || !count) return IA_E_ARG;
    if (n == 0) return hip_status(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    if (!dets || !keep || !workspace) return IA_E_ARG;
    size_t off[3];
    if (workspace_bytes < nms_workspace_bytes(1, n, 1, off)) return IA_E_WORKSPACE;
    char *ws = static_cast<char *>(workspace);
    uint64_t *adj = reinterpret_cast<uint64_t *>(ws + off[0]);
    uint16_t *sorted_rows = reinterpret_cast<uint16_t *>(ws + off[1]);
    int32_t *n_in = reinterpret_cast<int32_t *>(ws + off[2]);
    const int W = nms_adj_words(n);
    int rc = launch_adj(dets, 5, nullptr, 1, n, W, make_thr(iou_thr), 0.0f, adj, s);
    if (rc) return rc;
    SortSingleArgs sa;
    sa.dets = dets; sa.sorted_rows = sorted_rows; sa.n_in = n_in; sa.n = n;
    size_t lds = sizeof(uint64_t) * (size_t)pow2_at_least((uint32_t)n);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dets_sort),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_dets_sort, dim3(1), dim3(kSortThreads), lds, s, sa);
    if ((rc = hip_status(hipGetLastError()))) return rc;
    NmsSingleArgs a;
    a.sorted_rows = sorted_rows; a.n_in = n_in; a.adj = adj; a.keep = keep; a.count = count;
    a.n = n; a.W = W;
    hipLaunchKernelGGL(k_nms_single, dim3(1), dim3(kResThreads), 0, s, a);
    return hip_status(hipGetLastError());
}

// ------------------------------------------------------------------ per-image final top-k
// bbox_nms.py:49-56: concatenate kept boxes in class order; if more than
// max_num remain, sort by score descending and keep the first max_num
// (canonical tie order: position in the concatenation ascending).
//   k_final_keys  one workgroup per (image, class): 64-bit keys
//                 ordered(score)<<32 | ~position of its kept rows, written at the class's
//                 offset in the image's flat concatenation (coalesced);
//   k_finalize    one workgroup per image: radix top-k over the flat key array.
struct FinalKeyArgs {
    const float *scores_t;
    const int32_t *keep_count;
    const int32_t *keep_rows;
    uint64_t *flat;          // (B, C*Rs)
    int32_t Rs, C;
    const int32_t *gate;
};

__global__ void __launch_bounds__(256) k_final_keys(FinalKeyArgs a)
{
    __shared__ uint32_t s_prefix;
    const int c = blockIdx.x, b = blockIdx.y;
    if (a.gate && !a.gate[b]) return;
    const int32_t *kc = a.keep_count + (size_t)b * a.C;
    if (threadIdx.x < kWave) {
        uint32_t part = 0;
        for (int i = threadIdx.x; i < c; i += kWave) part += (uint32_t)kc[i];
        for (int off = 32; off > 0; off >>= 1) part += (uint32_t)__shfl_down((int)part, off);
        if (threadIdx.x == 0) s_prefix = part;
    }
    __syncthreads();
    const uint32_t prefix = s_prefix, n = (uint32_t)kc[c];
    const float *sc = a.scores_t + ((size_t)b * a.C + c) * a.Rs;
    const int32_t *kr = a.keep_rows + ((size_t)b * a.C + c) * a.Rs;
    uint64_t *out = a.flat + (size_t)b * a.C * a.Rs + prefix;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
        out[i] = ((uint64_t)ordered_key(sc[kr[i]]) << 32) | (uint64_t)(0xffffffffu - (prefix + i));
}

constexpr int kFinalParts = 16;      // workgroups per image for the first selection round

struct FinalArgs {
    const float *boxes;
    const float *scores_t;
    const int32_t *keep_count;
    const int32_t *keep_rows;
    const uint64_t *flat;
    uint64_t *part_keys;     // (B, kFinalParts, kpad): the parts' best max_per_img keys
    int32_t kpad;
    float *dets;
    int32_t *labels;
    int32_t *rows;
    int32_t *num;
    int32_t R, Rs, C, max_per_img;
    const int32_t *gate;
};

constexpr int kFinalThreads = 1024;
constexpr int kFinalMaxC = 1024;

// first round: each of kFinalParts workgroups keeps the best max_per_img keys of its slice of the
// image's flat key array (one workgroup over up to 375k keys was issue-bound in one CU)
__global__ void __launch_bounds__(kFinalThreads) k_finalize_part(FinalArgs a)
{
    __shared__ TopkScratch sc;
    __shared__ uint64_t sel[IA_MAX_PER_IMG];
    __shared__ uint32_t s_total;
    const int part = blockIdx.x, b = blockIdx.y;
    const uint32_t tid = threadIdx.x;
    if (a.gate && !a.gate[b]) return;
    if (tid < kWave) {
        uint32_t s = 0;
        for (int c = tid; c < a.C; c += kWave) s += (uint32_t)a.keep_count[(size_t)b * a.C + c];
        for (int off = 32; off > 0; off >>= 1) s += (uint32_t)__shfl_down((int)s, off);
        if (tid == 0) s_total = s;
    }
    __syncthreads();
    const uint32_t total = s_total, cap = (uint32_t)a.max_per_img;
    uint64_t *out = a.part_keys + ((size_t)b * kFinalParts + part) * a.kpad;
    if (total <= cap) return;                              // no selection needed at all
    const uint32_t chunk = (total + kFinalParts - 1) / kFinalParts;
    const uint32_t beg = part * chunk;
    const uint32_t cnt = (beg < total) ? ((total - beg < chunk) ? (total - beg) : chunk) : 0u;
    const uint64_t *flat = a.flat + (size_t)b * a.C * a.Rs + beg;
    const uint32_t kk = (cnt < cap) ? cnt : cap;
    if (kk == cnt) {
        for (uint32_t j = tid; j < (uint32_t)a.kpad; j += kFinalThreads) out[j] = (j < cnt) ? flat[j] : 0ull;
        return;
    }
    block_topk_desc([flat](uint32_t t) -> uint64_t { return flat[t]; }, cnt, kk, sc, sel);
    for (uint32_t j = tid; j < (uint32_t)a.kpad; j += kFinalThreads) out[j] = (j < kk) ? sel[j] : 0ull;
}

__global__ void __launch_bounds__(kFinalThreads) k_finalize(FinalArgs a)
{
    __shared__ TopkScratch sc;
    __shared__ uint64_t sel[IA_MAX_PER_IMG];
    __shared__ uint32_t prefix[kFinalMaxC + 1];
    const int b = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    if (a.gate && !a.gate[b]) return;
    const int C = a.C;
    const int32_t *kc = a.keep_count + (size_t)b * C;
    if (tid == 0) {
        uint32_t s = 0;
        for (int c = 0; c < C; ++c) { prefix[c] = s; s += (uint32_t)kc[c]; }
        prefix[C] = s;
    }
    __syncthreads();
    const uint32_t total = prefix[C];
    const uint32_t cap = (uint32_t)a.max_per_img;
    const uint32_t nd = (total < cap) ? total : cap;
    const float *sct = a.scores_t + (size_t)b * C * a.Rs;
    const int32_t *kr = a.keep_rows + (size_t)b * C * a.Rs;
    // position t in the concatenation -> (class, row)
    auto locate = [&](uint32_t t, int &c, int &r) {
        int lo = 0, hi = C;                     // largest c with prefix[c] <= t
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (prefix[mid] <= t) lo = mid; else hi = mid;
        }
        c = lo;
        r = kr[(size_t)c * a.Rs + (t - prefix[c])];
    };
    const bool need_sort = total > cap;
    if (need_sort && nd > 0) {
        // top-k of the union of the parts' survivors; unused slots are 0, below every real key
        const uint64_t *pk = a.part_keys + (size_t)b * kFinalParts * a.kpad;
        block_topk_desc([pk](uint32_t t) -> uint64_t { return pk[t]; },
                        (uint32_t)(kFinalParts * a.kpad), nd, sc, sel);
    }
    float *dets = a.dets + (size_t)b * a.max_per_img * 5;
    int32_t *labels = a.labels + (size_t)b * a.max_per_img;
    int32_t *rows = a.rows + (size_t)b * a.max_per_img;
    const float4 *bx = reinterpret_cast<const float4 *>(a.boxes) + (size_t)b * a.R;
    for (uint32_t d = tid; d < (uint32_t)a.max_per_img; d += kFinalThreads) {
        if (d < nd) {
            uint32_t t = need_sort ? (0xffffffffu - (uint32_t)sel[d]) : d;
            int c, r;
            locate(t, c, r);
            float4 q = bx[r];
            dets[5 * d + 0] = q.x; dets[5 * d + 1] = q.y; dets[5 * d + 2] = q.z;
            dets[5 * d + 3] = q.w; dets[5 * d + 4] = sct[(size_t)c * a.Rs + r];
            labels[d] = c; rows[d] = r;
        } else {
            for (int q = 0; q < 5; ++q) dets[5 * d + q] = 0.0f;
            labels[d] = -1; rows[d] = -1;
        }
    }
    if (tid == 0) a.num[b] = (int32_t)nd;
}

static int final_kpad() { return (IA_MAX_PER_IMG + 63) / 64 * 64; }

size_t finalize_workspace_bytes(int batch, int Rs, int C)
{
    return (size_t)batch * C * Rs * sizeof(uint64_t) +
           (size_t)batch * kFinalParts * final_kpad() * sizeof(uint64_t);
}

int launch_finalize(const float *boxes, const float *scores_t, const int32_t *keep_count,
                    const int32_t *keep_rows, int batch, int R, int Rs, int C, int max_per_img,
                    void *workspace, float *dets, int32_t *labels, int32_t *rows, int32_t *num,
                    hipStream_t s, const int32_t *gate)
{
    if (max_per_img > IA_MAX_PER_IMG) return IA_E_LIMIT_PER_IMG;
    if (batch < 1 || C < 1 || C > kFinalMaxC || max_per_img < 1)
        return IA_E_ARG;
    if (!dets || !labels || !rows || !num || !workspace) return IA_E_ARG;
    FinalKeyArgs k;
    k.gate = gate;
    k.scores_t = scores_t; k.keep_count = keep_count; k.keep_rows = keep_rows;
    k.flat = static_cast<uint64_t *>(workspace); k.Rs = Rs; k.C = C;
    hipLaunchKernelGGL(k_final_keys, dim3((unsigned)C, (unsigned)batch), dim3(256), 0, s, k);
    int rc = hip_status(hipGetLastError());
    if (rc) return rc;
    FinalArgs a;
    a.gate = gate;
    a.boxes = boxes; a.scores_t = scores_t; a.keep_count = keep_count; a.keep_rows = keep_rows;
    a.flat = k.flat; a.dets = dets; a.labels = labels; a.rows = rows; a.num = num;
    a.R = R; a.Rs = Rs; a.C = C; a.max_per_img = max_per_img;
    a.part_keys = k.flat + (size_t)batch * C * Rs;
    a.kpad = (max_per_img + 63) / 64 * 64;
    hipLaunchKernelGGL(k_finalize_part, dim3(kFinalParts, (unsigned)batch), dim3(kFinalThreads), 0, s, a);
    if ((rc = hip_status(hipGetLastError()))) return rc;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)batch), dim3(kFinalThreads), 0, s, a);
    return hip_status(hipGetLastError());
}

}  // namespace ia
