// Batched greedy NMS for all (image, class) problems in one launch, and the
// per-image final top-k.  Replaces multiclass_nms
// (reference mmdet/core/post_processing/bbox_nms.py:33-56) and the native op
// nms_cpu_kernel (reference mmdet/ops/nms/src/nms_cpu.cpp:4-59; ">=" at :55,
// ascending kept indices at :58).  The reference CUDA path (nms_kernel.cu)
// copies a bitmask to the host for every call; here everything stays on the
// device.
//
// One workgroup (16 wavefronts) per problem:
//   1. wave64-ballot compaction of the class column (score > score_thr) into
//      64-bit keys (ordered(score) << 32 | ~row), LDS bitonic sort ->
//      canonical order (score descending, row ascending);
//   2. every thread owns the boxes at sorted positions tid + i*1024 in
//      registers;
//   3. the sorted list is processed in chunks of 64: the owning wavefront
//      resolves the chunk with ballot masks (uniform loop over surviving
//      sources), publishes the survivors in LDS, and all wavefronts "push"
//      them onto their still-alive later candidates.  Work is
//      n * (kept + 64) pair tests instead of n^2/2, no mask matrix;
//   4. kept rows are emitted in ascending row order via an LDS bitmap.
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

namespace ia {

constexpr int kNmsThreads = 1024;
constexpr int kNmsWaves = kNmsThreads / kWave;
constexpr int kMaxOwn = IA_MAX_CANDIDATES / kNmsThreads;   // sorted positions owned per thread (max)

struct IouThr {
    double mid;      // midpoint between thr and its fp32 predecessor
    float thr;
    int32_t inclusive;
    int32_t is_half; // thr == 0.5f: fl(inter/uni) >= 0.5  <=>  2*inter >= uni, exact in fp32
};

static IouThr make_thr(float thr)
{
    IouThr t;
    t.thr = thr;
    float pred = nextafterf(thr, -INFINITY);
    t.mid = ((double)pred + (double)thr) * 0.5;
    uint32_t bits = __builtin_bit_cast(uint32_t, thr);
    t.inclusive = (bits & 1u) == 0u;
    t.is_half = (thr == 0.5f);
    return t;
}

// suppressor box s (earlier in the order, "i" of nms_cpu.cpp:37-55) vs candidate c ("j")
__device__ __forceinline__ bool suppresses(float sx1, float sy1, float sx2, float sy2, float sarea,
                                           float cx1, float cy1, float cx2, float cy2, float carea,
                                           const IouThr &t)
{
    float xx1 = (sx1 < cx1) ? cx1 : sx1;
    float yy1 = (sy1 < cy1) ? cy1 : sy1;
    float xx2 = (cx2 < sx2) ? cx2 : sx2;
    float yy2 = (cy2 < sy2) ? cy2 : sy2;
    float w = (xx2 - xx1) + 1.0f;  w = (0.0f < w) ? w : 0.0f;
    float h = (yy2 - yy1) + 1.0f;  h = (0.0f < h) ? h : 0.0f;
    float inter = w * h;
    float uni = (sarea + carea) - inter;
    if (uni > 0.0f) {
        // thr = 0.5 (every reference config): q = inter/uni rounds to >= 0.5 iff q >= 0.5 - 2^-26,
        // and no two fp32 numbers 2*inter < uni are closer than 2^-24 * uni, so the test is the
        // exact fp32 comparison 2*inter >= uni (2*inter is exact).
        if (t.is_half) return (inter + inter) >= uni;
        double lhs = (double)inter, rhs = t.mid * (double)uni;
        return t.inclusive ? (lhs >= rhs) : (lhs > rhs);
    }
    return (inter / uni) >= t.thr;
}

// value of lane `src` (wave-uniform index) in every lane: v_readlane_b32, no LDS round trip
__device__ __forceinline__ float bcast(float v, int src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

struct NmsSmem {
    float kx1[kWave], ky1[kWave], kx2[kWave], ky2[kWave], karea[kWave];
    uint32_t bitmap[IA_MAX_CANDIDATES / 32];
    uint32_t wave_cnt[kNmsWaves];
    uint32_t counter;
    uint32_t kept_n;
    uint32_t base;
};

// Greedy NMS of one problem by the whole workgroup.
//   R        rows in the problem's row space
//   Pred     row -> bool   (row takes part)
//   Score    row -> float
//   Box      row -> float4
// keep_out receives the kept rows ascending; returns the count (all threads).
template <int kOwn, class Pred, class Score, class Box>
__device__ uint32_t nms_block(uint32_t R, Pred pred, Score score, Box box, const IouThr &thr,
                              int32_t *keep_out, NmsSmem &sm, uint64_t *keys)
{
    const uint32_t tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    // 1. compaction
    if (tid == 0) sm.counter = 0;
    for (uint32_t i = tid; i < IA_MAX_CANDIDATES / 32; i += kNmsThreads) sm.bitmap[i] = 0;
    __syncthreads();
    const uint32_t R_up = (R + kWave - 1) & ~(uint32_t)(kWave - 1);
    for (uint32_t r = tid; r < R_up; r += kNmsThreads) {
        bool in = (r < R) && pred(r);
        uint64_t m = __ballot(in);
        if (m) {
            uint32_t base = 0;
            int leader = __builtin_ctzll(m);
            if (lane == leader) base = atomicAdd(&sm.counter, (uint32_t)__builtin_popcountll(m));
            base = (uint32_t)__shfl((int)base, leader);
            if (in)
                keys[base + lane_prefix_popc(m)] =
                    ((uint64_t)ordered_key(score(r)) << 32) | (uint64_t)(0xffffffffu - r);
        }
    }
    __syncthreads();
    const uint32_t n = sm.counter;
    if (n == 0) return 0;
    const uint32_t P = next_pow2(n < 2 ? 2 : n);
    for (uint32_t i = n + tid; i < P; i += kNmsThreads) keys[i] = 0;
    __syncthreads();
    bitonic_sort_desc(keys, P);

    // 2. ownership: sorted position tid + i*kNmsThreads
    float x1[kOwn], y1[kOwn], x2[kOwn], y2[kOwn], ar[kOwn];
    uint32_t row[kOwn];
    uint32_t alive = 0;
#pragma unroll
    for (int i = 0; i < kOwn; ++i) {
        uint32_t j = tid + i * kNmsThreads;
        x1[i] = y1[i] = x2[i] = y2[i] = ar[i] = 0.0f;
        row[i] = 0;
        if (j < n) {
            row[i] = 0xffffffffu - (uint32_t)keys[j];
            float4 bb = box(row[i]);
            x1[i] = bb.x; y1[i] = bb.y; x2[i] = bb.z; y2[i] = bb.w;
            ar[i] = ((bb.z - bb.x) + 1.0f) * ((bb.w - bb.y) + 1.0f);   // nms_cpu.cpp:18
            alive |= 1u << i;
        }
    }

    // 3. chunks of 64 sorted candidates
    const uint32_t nchunks = (n + kWave - 1) / kWave;
    for (uint32_t k = 0; k < nchunks; ++k) {
        const int owner = k % kNmsWaves, slot = k / kNmsWaves;
        if (wave == owner) {
            float bx1 = 0, by1 = 0, bx2 = 0, by2 = 0, bar = 0;
            uint32_t brow = 0;
            bool balive = false;
#pragma unroll
            for (int i = 0; i < kOwn; ++i)
                if (i == slot) {
                    bx1 = x1[i]; by1 = y1[i]; bx2 = x2[i]; by2 = y2[i]; bar = ar[i];
                    brow = row[i]; balive = (alive >> i) & 1u;
                }
            uint64_t amask = __ballot(balive);
            uint64_t todo = amask;
            while (todo) {
                const int src = __builtin_ctzll(todo);
                todo &= todo - 1;
                if (!((amask >> src) & 1ull)) continue;
                float sx1 = bcast(bx1, src), sy1 = bcast(by1, src);
                float sx2 = bcast(bx2, src), sy2 = bcast(by2, src), sar = bcast(bar, src);
                bool sup = (lane > src) && ((amask >> lane) & 1ull) &&
                           suppresses(sx1, sy1, sx2, sy2, sar, bx1, by1, bx2, by2, bar, thr);
                amask &= ~__ballot(sup);
            }
            const bool kept = (amask >> lane) & 1ull;
            if (kept) {
                uint32_t pos = lane_prefix_popc(amask);
                sm.kx1[pos] = bx1; sm.ky1[pos] = by1; sm.kx2[pos] = bx2; sm.ky2[pos] = by2;
                sm.karea[pos] = bar;
                atomicOr(&sm.bitmap[brow >> 5], 1u << (brow & 31u));
            }
            if (lane == 0) sm.kept_n = (uint32_t)__builtin_popcountll(amask);
        }
        __syncthreads();
        const uint32_t kn = sm.kept_n;
        const uint32_t chunk_end = (k + 1) * kWave;           // first position of later chunks
        if (kn > 0 && chunk_end < n) {
            for (uint32_t q = 0; q < kn; ++q) {
                const float sx1 = sm.kx1[q], sy1 = sm.ky1[q], sx2 = sm.kx2[q], sy2 = sm.ky2[q];
                const float sar = sm.karea[q];
#pragma unroll
                for (int i = 0; i < kOwn; ++i) {
                    if ((uint32_t)(i + 1) * kNmsThreads <= chunk_end) continue;   // uniform
                    uint32_t j = tid + i * kNmsThreads;
                    if (j >= chunk_end && ((alive >> i) & 1u) &&
                        suppresses(sx1, sy1, sx2, sy2, sar, x1[i], y1[i], x2[i], y2[i], ar[i], thr))
                        alive &= ~(1u << i);
                }
            }
        }
        __syncthreads();
    }

    // 4. emit kept rows ascending
    uint32_t total = 0;
    for (uint32_t r0 = 0; r0 < R_up; r0 += kNmsThreads) {
        uint32_t r = r0 + tid;
        bool kp = (r < R) && ((sm.bitmap[r >> 5] >> (r & 31u)) & 1u);
        uint64_t m = __ballot(kp);
        if (lane == 0) sm.wave_cnt[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kNmsWaves; ++w) {
            uint32_t c = sm.wave_cnt[w];
            before += (w < wave) ? c : 0;
            all += c;
        }
        if (kp) keep_out[total + before + lane_prefix_popc(m)] = (int32_t)r;
        total += all;
        __syncthreads();
    }
    return total;
}

// ------------------------------------------------------------------ per (image, class)
struct NmsArgs {
    const float *boxes;
    const float *scores_t;
    int32_t *keep_count;
    int32_t *keep_rows;
    IouThr thr;
    float score_thr;
    int32_t R, Rs, C;
};

template <int kOwn>
__global__ void __launch_bounds__(kNmsThreads) k_nms_class(NmsArgs a)
{
    extern __shared__ uint64_t keys[];
    __shared__ NmsSmem sm;
    const int c = blockIdx.x, b = blockIdx.y;
    const float *sc = a.scores_t + ((size_t)b * a.C + c) * a.Rs;
    const float4 *bx = reinterpret_cast<const float4 *>(a.boxes) + (size_t)b * a.R;
    const float st = a.score_thr;
    uint32_t cnt = nms_block<kOwn>((uint32_t)a.R,
                             [sc, st](uint32_t r) { return sc[r] > st; },      // bbox_nms.py:34
                             [sc](uint32_t r) { return sc[r]; },
                             [bx](uint32_t r) { return bx[r]; }, a.thr,
                             a.keep_rows + ((size_t)b * a.C + c) * a.Rs, sm, keys);
    if (threadIdx.x == 0) a.keep_count[(size_t)b * a.C + c] = (int32_t)cnt;
}

int launch_nms(const float *boxes, const float *scores_t, int batch, int R, int Rs, int C,
               float score_thr, float iou_thr, int32_t *keep_count, int32_t *keep_rows,
               hipStream_t s)
{
    if (batch < 1 || R < 1 || R > IA_MAX_CANDIDATES || C < 1 || Rs < R) return IA_E_ARG;
    if (!boxes || !scores_t || !keep_count || !keep_rows) return IA_E_ARG;
    NmsArgs a;
    a.boxes = boxes; a.scores_t = scores_t; a.keep_count = keep_count; a.keep_rows = keep_rows;
    a.thr = make_thr(iou_thr); a.score_thr = score_thr; a.R = R; a.Rs = Rs; a.C = C;
    uint32_t P = 2;
    while (P < (uint32_t)R) P <<= 1;
    size_t lds = sizeof(uint64_t) * P;
    const dim3 grid((unsigned)C, (unsigned)batch), block(kNmsThreads);
#define IA_LAUNCH_NMS(OWN)                                                                        \
    do {                                                                                          \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_nms_class<OWN>),      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                       \
        hipLaunchKernelGGL(k_nms_class<OWN>, grid, block, lds, s, a);                             \
    } while (0)
    const int own = (R + kNmsThreads - 1) / kNmsThreads;    // registers follow the real row count
    if (own <= 1) IA_LAUNCH_NMS(1);
    else if (own <= 2) IA_LAUNCH_NMS(2);
    else if (own <= 3) IA_LAUNCH_NMS(3);
    else if (own <= 5) IA_LAUNCH_NMS(5);
    else IA_LAUNCH_NMS(kMaxOwn);
#undef IA_LAUNCH_NMS
    return hip_status(hipGetLastError());
}

// ------------------------------------------------------------------ standalone op
struct NmsSingleArgs {
    const float *dets;
    int32_t *keep;
    int32_t *count;
    IouThr thr;
    int32_t n;
};

__global__ void __launch_bounds__(kNmsThreads) k_nms_single(NmsSingleArgs a)
{
    extern __shared__ uint64_t keys[];
    __shared__ NmsSmem sm;
    const float *d = a.dets;
    uint32_t cnt = nms_block<kMaxOwn>((uint32_t)a.n, [](uint32_t) { return true; },
                             [d](uint32_t r) { return d[5 * (size_t)r + 4]; },
                             [d](uint32_t r) {
                                 const float *q = d + 5 * (size_t)r;
                                 return make_float4(q[0], q[1], q[2], q[3]);
                             },
                             a.thr, a.keep, sm, keys);
    if (threadIdx.x == 0) *a.count = (int32_t)cnt;
}

int launch_nms_single(const float *dets, int n, float iou_thr, int32_t *keep, int32_t *count,
                      hipStream_t s)
{
    if (n < 0 || n > IA_MAX_CANDIDATES || !count) return IA_E_ARG;
    if (n == 0) return hip_status(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    if (!dets || !keep) return IA_E_ARG;
    NmsSingleArgs a;
    a.dets = dets; a.keep = keep; a.count = count; a.thr = make_thr(iou_thr); a.n = n;
    uint32_t P = 2;
    while (P < (uint32_t)n) P <<= 1;
    size_t lds = sizeof(uint64_t) * P;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_nms_single),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_nms_single, dim3(1), dim3(kNmsThreads), lds, s, a);
    return hip_status(hipGetLastError());
}

// ------------------------------------------------------------------ per-image final top-k
// bbox_nms.py:49-56: concatenate kept boxes in class order; if more than
// max_num remain, sort by score descending and keep the first max_num
// (canonical tie order: position in the concatenation ascending).
struct FinalArgs {
    const float *boxes;
    const float *scores_t;
    const int32_t *keep_count;
    const int32_t *keep_rows;
    float *dets;
    int32_t *labels;
    int32_t *rows;
    int32_t *num;
    int32_t R, Rs, C, max_per_img;
};

constexpr int kFinalThreads = 1024;
constexpr int kFinalMaxC = 1024;

__global__ void __launch_bounds__(kFinalThreads) k_finalize(FinalArgs a)
{
    __shared__ TopkScratch sc;
    __shared__ uint64_t sel[IA_MAX_PER_IMG];
    __shared__ uint32_t prefix[kFinalMaxC + 1];
    const int b = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const int C = a.C;
    const int32_t *kc = a.keep_count + (size_t)b * C;
    if (tid == 0) {
        uint32_t s = 0;
        for (int c = 0; c < C; ++c) { prefix[c] = s; s += (uint32_t)kc[c]; }
        prefix[C] = s;
    }
    __syncthreads();
    const uint32_t total = prefix[C];
    const uint32_t cap = (a.max_per_img < 0) ? total : (uint32_t)a.max_per_img;
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
        auto key = [&](uint32_t t) -> uint64_t {
            int c, r;
            locate(t, c, r);
            return ((uint64_t)ordered_key(sct[(size_t)c * a.Rs + r]) << 32) |
                   (uint64_t)(0xffffffffu - t);
        };
        block_topk_desc(key, total, nd, sc, sel);
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

int launch_finalize(const float *boxes, const float *scores_t, const int32_t *keep_count,
                    const int32_t *keep_rows, int batch, int R, int Rs, int C, int max_per_img,
                    float *dets, int32_t *labels, int32_t *rows, int32_t *num, hipStream_t s)
{
    if (batch < 1 || C < 1 || C > kFinalMaxC || max_per_img < 1 || max_per_img > IA_MAX_PER_IMG)
        return IA_E_ARG;
    if (!dets || !labels || !rows || !num) return IA_E_ARG;
    FinalArgs a;
    a.boxes = boxes; a.scores_t = scores_t; a.keep_count = keep_count; a.keep_rows = keep_rows;
    a.dets = dets; a.labels = labels; a.rows = rows; a.num = num;
    a.R = R; a.Rs = Rs; a.C = C; a.max_per_img = max_per_img;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)batch), dim3(kFinalThreads), 0, s, a);
    return hip_status(hipGetLastError());
}

}  // namespace ia
