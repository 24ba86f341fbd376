// Lazy multiclass NMS: the per-image top max_per_img detections WITHOUT resolving the 80 class
// problems completely (reference mmdet/core/post_processing/bbox_nms.py:33-56 +
// mmdet/ops/nms/src/nms_cpu.cpp:4-59 -- same output, different evaluation order).
//
// multiclass_nms runs greedy NMS per class and then keeps the max_per_img best-scored survivors
// of all classes.  Greedy NMS decides a box only from HIGHER-ranked boxes of its own class, and
// the final selection takes the survivors in (score desc, class asc, row asc) order -- which
// inside a class is exactly the NMS order.  So walking ALL (class, box) pairs of an image in
// that one global order and keeping a pair iff no already kept pair of the same class suppresses
// it yields the survivors in final order; after max_per_img + 1 survivors nothing further can
// matter (the extra one only proves that more than max_num boxes survive, i.e. that the reference
// sorts by score, bbox_nms.py:52-56; with at most max_num survivors it keeps the concatenation
// order, which the walk restores from the pair indices).
// With random-init weights (every one of the 375 440 pairs passes score_thr) ~220 pairs are
// touched per image instead of 4 693 per class x 80 classes.
//
//   k_lazy_keys     one workgroup per (image, class): keys ordered(score) << 32 | ~(c*Rs + r) of
//                   the pairs with score > score_thr, compacted into the image's flat key array;
//   k_lazy_part     16 workgroups per image: each keeps the best M keys of its slice;
//   k_lazy_greedy   one workgroup per image: best M of the union (radix select + sort in LDS), the
//                   boxes of those M pairs staged in LDS, then one wavefront walks them 64 at a
//                   time: lane = candidate, tested against the kept list (LDS broadcast reads)
//                   and, inside the chunk, against earlier unsuppressed lanes (readlane);
//                   survivors are the detections, in order.
// If an image runs out of its M candidates with fewer than max_per_img survivors while more
// pairs exist, need_full[b] is set and the complete path (k_adj / k_class_sort / k_nms_resolve /
// k_finalize, gated by that flag) produces that image's result instead.
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_block.hpp"
#include "ia_nms.hpp"

namespace ia {

constexpr int kLazyMax = 4096;        // candidates walked per image at most (LDS: 133 KB of 160 KB)
constexpr int kLazyFirst = 1024;      // first, cheap walk of the default two-tier schedule
constexpr int kLazyParts = 16;
constexpr int kLazyThreads = 1024;

struct LazyArgs {
    const float *boxes;               // (B, R, 4)
    const float *scores_t;            // (B, C, Rs)
    uint64_t *flat;                   // (B, C*Rs) compacted keys
    uint64_t *part_keys;              // (B, kLazyParts, M)
    int32_t *count;                   // (B) pairs above score_thr
    float *dets;
    int32_t *labels, *rows, *num, *need_full;
    IouThr thr;
    float score_thr;
    int32_t R, Rs, C, M, max_per_img;
    const int32_t *gate;              // optional (B): second, larger walk only where the first failed
};

__global__ void __launch_bounds__(256) k_lazy_keys(LazyArgs a)
{
    __shared__ uint32_t s_base, s_cnt;
    const int c = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & (kWave - 1);
    const float *sc = a.scores_t + ((size_t)b * a.C + c) * a.Rs;
    uint64_t *out = a.flat + (size_t)b * a.C * a.Rs;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    // two passes over the class column (L2-resident): count, reserve a range, write
    uint32_t mine = 0;
    for (int r0 = 0; r0 < a.R; r0 += 256) {
        const int r = r0 + threadIdx.x;
        mine += (r < a.R && sc[r] > a.score_thr) ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) mine += (uint32_t)__shfl_down((int)mine, off);
    if (lane == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        s_base = s_cnt ? (uint32_t)atomicAdd(a.count + b, (int32_t)s_cnt) : 0u;
        s_cnt = 0;
    }
    __syncthreads();
    const uint32_t base = s_base;
    for (int r0 = 0; r0 < a.R; r0 += 256) {
        const int r = r0 + threadIdx.x;
        const bool in = r < a.R && sc[r] > a.score_thr;                       // bbox_nms.py:34
        const uint64_t m = __ballot(in);
        if (m) {
            uint32_t w0 = 0;
            const int leader = __builtin_ctzll(m);
            if (lane == leader) w0 = atomicAdd(&s_cnt, (uint32_t)__builtin_popcountll(m));
            w0 = (uint32_t)__shfl((int)w0, leader);
            if (in)
                out[base + w0 + lane_prefix_popc(m)] =
                    ((uint64_t)ordered_key(sc[r]) << 32) |
                    (uint64_t)(0xffffffffu - (uint32_t)(c * a.Rs + r));
        }
    }
}

__global__ void __launch_bounds__(kLazyThreads) k_lazy_part(LazyArgs a)
{
    __shared__ TopkScratch sc;
    __shared__ uint64_t sel[kLazyMax];
    const int part = blockIdx.x, b = blockIdx.y;
    const uint32_t tid = threadIdx.x;
    if (a.gate && !a.gate[b]) return;
    const uint32_t total = (uint32_t)a.count[b], M = (uint32_t)a.M;
    if (total <= M) return;                                // the greedy kernel reads `flat` directly
    uint64_t *out = a.part_keys + ((size_t)b * kLazyParts + part) * M;
    const uint32_t chunk = (total + kLazyParts - 1) / kLazyParts;
    const uint32_t beg = part * chunk;
    const uint32_t cnt = (beg < total) ? ((total - beg < chunk) ? (total - beg) : chunk) : 0u;
    const uint64_t *flat = a.flat + (size_t)b * a.C * a.Rs + beg;
    const uint32_t kk = (cnt < M) ? cnt : M;
    if (kk == cnt) {
        for (uint32_t j = tid; j < M; j += kLazyThreads) out[j] = (j < cnt) ? flat[j] : 0ull;
        return;
    }
    block_topk_desc([flat](uint32_t t) -> uint64_t { return flat[t]; }, cnt, kk, sc, sel);
    for (uint32_t j = tid; j < M; j += kLazyThreads) out[j] = (j < kk) ? sel[j] : 0ull;
}

struct LazySmem {
    float4 box[kLazyMax];                 // boxes of the candidates, in order
    float4 kbox[IA_MAX_PER_IMG + 1];      // kept (one more than max_per_img, see below): box,
    float karea[IA_MAX_PER_IMG + 1];      // area, class, pair index c*Rs + r, score
    int32_t kcls[IA_MAX_PER_IMG + 1];
    uint32_t kidx[IA_MAX_PER_IMG + 1];
    float kscore[IA_MAX_PER_IMG + 1];
};

__global__ void __launch_bounds__(kLazyThreads) k_lazy_greedy(LazyArgs a)
{
    __shared__ TopkScratch sc;
    __shared__ uint64_t sel[kLazyMax];
    extern __shared__ unsigned char dyn[];
    LazySmem &sm = *reinterpret_cast<LazySmem *>(dyn);
    const int b = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    if (a.gate && !a.gate[b]) return;
    const uint32_t total = (uint32_t)a.count[b], M = (uint32_t)a.M;
    const uint32_t m = (total < M) ? total : M;
    // ---- the m globally best pairs of the image, sorted (score desc, class asc, row asc)
    if (total > M) {
        const uint64_t *pk = a.part_keys + (size_t)b * kLazyParts * M;
        block_topk_desc([pk](uint32_t t) -> uint64_t { return pk[t]; }, kLazyParts * M, m, sc, sel);
    } else if (m > 0) {
        const uint64_t *flat = a.flat + (size_t)b * a.C * a.Rs;
        const uint32_t P = next_pow2(m < 2 ? 2 : m);
        for (uint32_t i = tid; i < P; i += kLazyThreads) sel[i] = (i < m) ? flat[i] : 0ull;
        __syncthreads();
        bitonic_sort_desc(sel, P);
    }
    __syncthreads();
    const float4 *bx = reinterpret_cast<const float4 *>(a.boxes) + (size_t)b * a.R;
    for (uint32_t i = tid; i < m; i += kLazyThreads) {
        const uint32_t idx = 0xffffffffu - (uint32_t)sel[i];
        sm.box[i] = bx[idx % (uint32_t)a.Rs];
    }
    __syncthreads();
    float *dets = a.dets + (size_t)b * a.max_per_img * 5;
    int32_t *labels = a.labels + (size_t)b * a.max_per_img;
    int32_t *rows = a.rows + (size_t)b * a.max_per_img;
    if (tid < kWave) {
        const int lane = tid;
        const uint32_t cap = (uint32_t)a.max_per_img;
        // bbox_nms.py:52-56 sorts by score ONLY when more than max_num boxes survive; with at
        // most max_num survivors the result keeps the concatenation order (class asc, row asc).
        // So the walk goes on until survivor number cap + 1 proves that the reference sorts.
        const uint32_t want = cap + 1;
        uint32_t nk = 0;
        for (uint32_t i0 = 0; i0 < m && nk < want; i0 += kWave) {
            const uint32_t i = i0 + lane;
            const bool live = i < m;
            const uint64_t key = live ? sel[i] : 0ull;
            const uint32_t idx = 0xffffffffu - (uint32_t)key;
            const int cls = live ? (int)(idx / (uint32_t)a.Rs) : -1;
            const float4 q = live ? sm.box[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float area = ((q.z - q.x) + 1.0f) * ((q.w - q.y) + 1.0f);      // nms_cpu.cpp:18
            // (A) against the pairs kept in earlier chunks
            bool sup = !live;
            for (uint32_t j = 0; j < nk; ++j) {
                const float4 k = sm.kbox[j];                                      // LDS broadcast
                if (!sup && sm.kcls[j] == cls &&
                    suppresses(k.x, k.y, k.z, k.w, sm.karea[j], q.x, q.y, q.z, q.w, area, a.thr))
                    sup = true;
            }
            // (B) inside the chunk, in order: an unsuppressed lane is kept and suppresses later lanes
            uint64_t alive = __ballot(!sup);
            while (alive && nk < want) {
                const int j = __builtin_ctzll(alive);
                alive &= alive - 1;
                const float jx1 = __shfl(q.x, j), jy1 = __shfl(q.y, j), jx2 = __shfl(q.z, j);
                const float jy2 = __shfl(q.w, j), jar = __shfl(area, j);
                const int jc = __shfl(cls, j);
                if (lane == j) {
                    sm.kbox[nk] = q; sm.karea[nk] = area; sm.kcls[nk] = cls; sm.kidx[nk] = idx;
                    sm.kscore[nk] = ordered_key_inv((uint32_t)(key >> 32));
                }
                ++nk;
                const bool hit = lane > j && !sup && cls == jc &&
                                 suppresses(jx1, jy1, jx2, jy2, jar, q.x, q.y, q.z, q.w, area, a.thr);
                if (hit) sup = true;
                alive &= ~__ballot(hit);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const bool sorted = nk > cap;                        // more than max_num survive: score order
        const bool all_seen = !sorted && total <= M;         // every pair was walked: that is all
        const bool done = sorted || all_seen;
        const uint32_t nout = sorted ? cap : nk;
        if (lane == 0) {
            a.need_full[b] = done ? 0 : 1;
            if (done) a.num[b] = (int32_t)nout;
        }
        if (done) {
            for (uint32_t e = lane; e < nout; e += kWave) {
                uint32_t pos = e;
                if (!sorted) {                               // concatenation order = pair index order
                    pos = 0;
                    const uint32_t me = sm.kidx[e];
                    for (uint32_t j = 0; j < nout; ++j) pos += (sm.kidx[j] < me) ? 1u : 0u;
                }
                const float4 q = sm.kbox[e];
                dets[5 * pos + 0] = q.x; dets[5 * pos + 1] = q.y; dets[5 * pos + 2] = q.z;
                dets[5 * pos + 3] = q.w; dets[5 * pos + 4] = sm.kscore[e];
                labels[pos] = sm.kcls[e];
                rows[pos] = (int32_t)(sm.kidx[e] % (uint32_t)a.Rs);
            }
            for (uint32_t d = nout + lane; d < cap; d += kWave) {
                for (int q5 = 0; q5 < 5; ++q5) dets[5 * d + q5] = 0.0f;
                labels[d] = -1; rows[d] = -1;
            }
        }
    }
}

static size_t lazy_align(size_t v) { return (v + 255) / 256 * 256; }

size_t lazy_workspace_bytes(int batch, int Rs, int C)
{
    return lazy_align((size_t)batch * C * Rs * sizeof(uint64_t)) +
           lazy_align((size_t)batch * kLazyParts * kLazyMax * sizeof(uint64_t)) +
           lazy_align((size_t)batch * sizeof(int32_t));
}

int launch_lazy_nms(const float *boxes, const float *scores_t, int batch, int R, int Rs, int C,
                    float score_thr, float iou_thr, int max_per_img, int candidates,
                    void *workspace, float *dets, int32_t *labels, int32_t *rows, int32_t *num,
                    int32_t *need_full, hipStream_t s)
{
    if (R > IA_MAX_CANDIDATES) return IA_E_LIMIT_BOXES;
    if (max_per_img > IA_MAX_PER_IMG) return IA_E_LIMIT_PER_IMG;
    if (batch < 1 || R < 1 || Rs < R || C < 1 || max_per_img < 1)
        return IA_E_ARG;
    if ((uint64_t)C * (uint64_t)Rs > 0x7fffffffull) return IA_E_ARG;
    if (!boxes || !scores_t || !workspace || !dets || !labels || !rows || !num || !need_full)
        return IA_E_ARG;
    // candidates == 0: two tiers -- a short walk (kLazyFirst pairs, enough for most images) and,
    // only for the images it could not finish, a long one (kLazyMax) before the complete path
    const bool two_tier = candidates <= 0 && max_per_img < kLazyFirst;
    if (candidates <= 0) candidates = two_tier ? kLazyFirst : kLazyMax;
    if (candidates > kLazyMax) candidates = kLazyMax;
    if (candidates <= max_per_img) candidates = max_per_img + 1;   // the walk needs cap + 1 survivors
    if (candidates > kLazyMax) candidates = kLazyMax;
    LazyArgs a;
    a.gate = nullptr;
    char *ws = static_cast<char *>(workspace);
    a.flat = reinterpret_cast<uint64_t *>(ws);
    ws += lazy_align((size_t)batch * C * Rs * sizeof(uint64_t));
    a.part_keys = reinterpret_cast<uint64_t *>(ws);
    ws += lazy_align((size_t)batch * kLazyParts * kLazyMax * sizeof(uint64_t));
    a.count = reinterpret_cast<int32_t *>(ws);
    a.boxes = boxes; a.scores_t = scores_t; a.dets = dets; a.labels = labels; a.rows = rows;
    a.num = num; a.need_full = need_full; a.thr = make_thr(iou_thr); a.score_thr = score_thr;
    a.R = R; a.Rs = Rs; a.C = C; a.M = candidates; a.max_per_img = max_per_img;
    hipError_t e = hipMemsetAsync(a.count, 0, sizeof(int32_t) * (size_t)batch, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_lazy_keys, dim3((unsigned)C, (unsigned)batch), dim3(256), 0, s, a);
    int rc = hip_status(hipGetLastError());
    if (rc) return rc;
    hipLaunchKernelGGL(k_lazy_part, dim3(kLazyParts, (unsigned)batch), dim3(kLazyThreads), 0, s, a);
    if ((rc = hip_status(hipGetLastError()))) return rc;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_lazy_greedy),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LazySmem));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_lazy_greedy, dim3((unsigned)batch), dim3(kLazyThreads), sizeof(LazySmem), s, a);
    if ((rc = hip_status(hipGetLastError())) || !two_tier) return rc;
    a.M = kLazyMax;
    a.gate = need_full;               // read at kernel start, rewritten by the same workgroup at its end
    hipLaunchKernelGGL(k_lazy_part, dim3(kLazyParts, (unsigned)batch), dim3(kLazyThreads), 0, s, a);
    if ((rc = hip_status(hipGetLastError()))) return rc;
    hipLaunchKernelGGL(k_lazy_greedy, dim3((unsigned)batch), dim3(kLazyThreads), sizeof(LazySmem), s, a);
    return hip_status(hipGetLastError());
}

}  // namespace ia
