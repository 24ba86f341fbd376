// soft-NMS: mmdet/ops/nms/src/soft_nms_cpu.pyx:22-127 behind
// mmdet/ops/nms/nms_wrapper.py:52-78, and multiclass_nms with nms.type='soft_nms'
// (mmdet/core/post_processing/bbox_nms.py:29-56).
//
// The algorithm is sequential by definition (n selections, each re-weights the rest), so the
// parallelism is (image x class) problems x the positions of one problem: one 256-thread
// workgroup per problem, ONE barrier per selection:
//   * the reference's working array is emulated position by position (swap of the maximum into
//     slot i, discard by moving the last box into the hole), because ties resolve by position
//     (:52-56 strict `<` scan => the first position wins) and the output order is the slot order;
//   * slot p is owned by thread p mod 256 for the whole run: (score, element) of a slot live in
//     LDS and are only ever touched by their owner, except slot i which the owner of `maxpos`
//     reads once (the swap never has to be written to slot i: the selected box goes straight to
//     the output);
//   * the pass that re-weights the slots (i, N) also finds the next maximum: 64-bit key
//     ordered(score) << 32 | ~pos << 16 | element, wave reduction, the lane that holds its wave's
//     best publishes the box, and `__syncthreads_or(discarded)` is both the barrier of the
//     reduction and the test for the rare compaction step;
//   * weights follow the mixed fp32 / fp64 arithmetic of the Cython-generated C literally (see
//     oracle/iouaware_oracle_softnms.c) -> same bits as the reference module.
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_block.hpp"

namespace ia {

constexpr int kSoftThreads = 256;
constexpr int kSoftWaves = kSoftThreads / kWave;
constexpr uint16_t kDeadBit = 0x8000;

struct SoftParams { float iou_thr, sigma, min_score; int32_t method; };

__device__ __forceinline__ float sn_max(float a, float b) { return (a >= b) ? a : b; }   // :15-16
__device__ __forceinline__ float sn_min(float a, float b) { return (a <= b) ? a : b; }   // :18-19

// :79-105 for one (selected box t, box q) pair; false when the boxes do not overlap (the score
// is then neither re-weighted nor tested against min_score)
__device__ __forceinline__ bool soft_weight(const float4 &t, const float4 &q, const SoftParams &sp,
                                            float &weight)
{
    const float area = (float)(((double)(q.z - q.x) + 1.0) * ((double)(q.w - q.y) + 1.0));
    const float iw = (float)((double)(sn_min(t.z, q.z) - sn_max(t.x, q.x)) + 1.0);
    if (!(iw > 0.0f)) return false;
    const float ih = (float)((double)(sn_min(t.w, q.w) - sn_max(t.y, q.y)) + 1.0);
    if (!(ih > 0.0f)) return false;
    const float inter = iw * ih;
    const float ua = (float)(((((double)(t.z - t.x) + 1.0) * ((double)(t.w - t.y) + 1.0)) +
                              (double)area) - (double)inter);
    const float ov = inter / ua;
    if (sp.method == 1)      weight = (ov > sp.iou_thr) ? (float)(1.0 - (double)ov) : 1.0f;
    else if (sp.method == 2) weight = (float)exp_f64_((double)((-(ov * ov)) / sp.sigma));
    else                     weight = (ov > sp.iou_thr) ? 0.0f : 1.0f;
    return true;
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask)
{
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask);
    return ((uint64_t)hi << 32) | lo;
}

struct SoftSmem {
    uint64_t red[2][kSoftWaves];
    float4 box[2][kSoftWaves];
    uint32_t wcnt[kSoftWaves];
    uint32_t n_new, k_slots;
};

// Src: pred(r) / score(r) / box(r) over the R input rows; Out: emit(slot, row, score), done(N)
template <class Src, class Out>
__device__ void soft_nms_block(uint32_t R, const Src &src, Out &out, const SoftParams &sp,
                               float *s_score, uint16_t *s_elem, uint16_t *s_list, SoftSmem &sm)
{
    const uint32_t tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave;

    // ---- the problem's boxes in input order (order-preserving compaction, bbox_nms.py:34-44)
    uint32_t N = 0;
    for (uint32_t base = 0; base < R; base += kSoftThreads) {
        const uint32_t r = base + tid;
        const bool in = (r < R) && src.pred(r);
        const uint64_t m = __ballot(in);
        if (lane == 0) sm.wcnt[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kSoftWaves; ++w) {
            const uint32_t c = sm.wcnt[w];
            before += (w < wave) ? c : 0u;
            all += c;
        }
        if (in) {
            const uint32_t p = N + before + lane_prefix_popc(m);
            s_score[p] = src.score(r);
            s_elem[p] = (uint16_t)r;
        }
        N += all;
        __syncthreads();
    }

    // key of slot p: larger score first, then the lower position; the element rides along
    auto make_key = [](float s, uint32_t p, uint32_t e) -> uint64_t {
        return ((uint64_t)ordered_key(s) << 32) | ((uint64_t)(0xffffu - p) << 16) | e;
    };
    // wave-reduce (key, box) and publish the wave's best in buffer `par`
    auto publish = [&](uint64_t key, const float4 &bx, int par) {
        uint64_t best = key;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint64_t o = shfl_xor_u64(best, off);
            best = (o > best) ? o : best;
        }
        if (key == best && (key != 0 || lane == 0)) {      // keys are unique unless all are 0
            sm.red[par][wave] = best;
            sm.box[par][wave] = bx;
        }
    };
    // full scan of the slots [from, N): used at the start and after a compaction
    auto scan = [&](uint32_t from, int par) {
        uint64_t key = 0;
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t p = from + ((tid + kSoftThreads - (from % kSoftThreads)) % kSoftThreads);
        for (; p < N; p += kSoftThreads) {
            const uint32_t e = s_elem[p];
            const uint64_t k = make_key(s_score[p], p, e);
            if (k > key) { key = k; bx = src.box(e); }
        }
        publish(key, bx, par);
    };

    int par = 0;
    scan(0, par);
    __syncthreads();
    for (uint32_t i = 0; i < N; ++i) {
        // ---- the winner of the slots [i, N)
        uint64_t best = 0;
        int bw = 0;
#pragma unroll
        for (int w = 0; w < kSoftWaves; ++w) {
            const uint64_t k = sm.red[par][w];
            if (k > best) { best = k; bw = w; }
        }
        const float4 t = sm.box[par][bw];
        const float ts = ordered_key_inv((uint32_t)(best >> 32));
        const uint32_t maxpos = 0xffffu - (uint32_t)((best >> 16) & 0xffffu);
        const uint32_t te = (uint32_t)(best & 0xffffu);
        if (tid == 0) out.emit(i, te, ts, t);
        par ^= 1;
        // ---- re-weight the slots (i, N); the owner of `maxpos` takes over the box of slot i
        uint64_t key = 0;
        float4 kbx = make_float4(0.f, 0.f, 0.f, 0.f);
        bool dead = false;
        uint32_t p = (i + 1) + ((tid + kSoftThreads - ((i + 1) % kSoftThreads)) % kSoftThreads);
        for (; p < N; p += kSoftThreads) {
            uint32_t e;
            float s;
            if (p == maxpos) { e = s_elem[i]; s = s_score[i]; }
            else             { e = s_elem[p]; s = s_score[p]; }
            const float4 q = src.box(e);
            float w;
            bool d = false;
            if (soft_weight(t, q, sp, w)) {
                s = w * s;
                d = s < sp.min_score;                       // :113
            }
            s_score[p] = s;
            s_elem[p] = (uint16_t)(e | (d ? kDeadBit : 0));
            dead |= d;
            if (!d) {
                const uint64_t k = make_key(s, p, e);
                if (k > key) { key = k; kbx = q; }
            }
        }
        publish(key, kbx, par);
        if (__syncthreads_or(dead ? 1 : 0)) {
            // ---- discards (:113-122): the holes below the new N take the live boxes from the
            // top, highest position first -- what the sequential "move the last box here" does
            if (wave == 0) {
                uint32_t nd = 0;
                for (uint32_t base = i + 1; base < N; base += kWave) {
                    const uint32_t q = base + lane;
                    nd += (uint32_t)__builtin_popcountll(__ballot(q < N && (s_elem[q] & kDeadBit)));
                }
                const uint32_t Nn = N - nd;
                uint32_t K = 0;
                for (uint32_t base = i + 1; base < Nn; base += kWave) {
                    const uint32_t q = base + lane;
                    const bool h = q < Nn && (s_elem[q] & kDeadBit);
                    const uint64_t m = __ballot(h);
                    if (h) s_list[K + lane_prefix_popc(m)] = (uint16_t)q;
                    K += (uint32_t)__builtin_popcountll(m);
                }
                if (lane == 0) { sm.n_new = Nn; sm.k_slots = K; }
            }
            __syncthreads();
            const uint32_t Nn = sm.n_new;
            if (wave == 0) {
                uint32_t K2 = 0;
                for (uint32_t top = N; top > Nn; top = (top > kWave) ? top - kWave : 0) {
                    const bool ok = top > (uint32_t)lane;
                    const uint32_t q = ok ? top - 1 - lane : 0;
                    const bool lv = ok && q >= Nn && !(s_elem[q] & kDeadBit);
                    const uint64_t m = __ballot(lv);
                    if (lv) {
                        const uint32_t dst = s_list[K2 + lane_prefix_popc(m)];
                        s_score[dst] = s_score[q];
                        s_elem[dst] = s_elem[q];
                    }
                    K2 += (uint32_t)__builtin_popcountll(m);
                    if (top <= kWave) break;
                }
            }
            __syncthreads();
            N = Nn;
            scan(i + 1, par);
            __syncthreads();
        }
    }
    if (tid == 0) out.done(N);
}

// ------------------------------------------------------------------ multiclass (get_bboxes)
struct SoftClassSrc {
    const float *sc;               // class column of scores_t
    const float4 *bx;              // (R) boxes of the image
    float thr;
    __device__ __forceinline__ bool pred(uint32_t r) const { return sc[r] > thr; }
    __device__ __forceinline__ float score(uint32_t r) const { return sc[r]; }
    __device__ __forceinline__ float4 box(uint32_t r) const { return bx[r]; }
};
struct SoftClassOut {
    int32_t *keep_rows;            // selection order
    float *soft_scores;            // decayed score BY ROW (what the final top-k reads)
    int32_t *keep_count;
    __device__ __forceinline__ void emit(uint32_t slot, uint32_t row, float s, const float4 &)
    {
        keep_rows[slot] = (int32_t)row;
        soft_scores[row] = s;
    }
    __device__ __forceinline__ void done(uint32_t n) { *keep_count = (int32_t)n; }
};

struct SoftArgs {
    const float *boxes, *scores_t;
    int32_t *keep_count, *keep_rows;
    float *soft_scores;
    float score_thr;
    SoftParams sp;
    int32_t R, Rs, C, cap;
};

__global__ void __launch_bounds__(kSoftThreads) k_soft_nms(SoftArgs a)
{
    extern __shared__ float s_dyn[];
    __shared__ SoftSmem sm;
    float *s_score = s_dyn;
    uint16_t *s_elem = reinterpret_cast<uint16_t *>(s_dyn + a.cap);
    uint16_t *s_list = s_elem + a.cap;
    const int c = blockIdx.x, b = blockIdx.y;
    const size_t prob = (size_t)b * a.C + c;
    SoftClassSrc src{a.scores_t + prob * a.Rs,
                     reinterpret_cast<const float4 *>(a.boxes) + (size_t)b * a.R, a.score_thr};
    SoftClassOut out{a.keep_rows + prob * a.Rs, a.soft_scores + prob * a.Rs, a.keep_count + prob};
    soft_nms_block((uint32_t)a.R, src, out, a.sp, s_score, s_elem, s_list, sm);
}

static int soft_cap(int R) { return (R + 63) / 64 * 64; }
static size_t soft_lds_bytes(int cap) { return (size_t)cap * (sizeof(float) + 2 * sizeof(uint16_t)); }

int launch_soft_nms(const float *boxes, const float *scores_t, int batch, int R, int Rs, int C,
                    float score_thr, float iou_thr, int method, float sigma, float min_score,
                    int32_t *keep_count, int32_t *keep_rows, float *soft_scores, hipStream_t s)
{
    if (R > IA_MAX_CANDIDATES) return IA_E_LIMIT_BOXES;
    if (batch < 1 || R < 1 || Rs < R || C < 1) return IA_E_ARG;
    if (!boxes || !scores_t || !keep_count || !keep_rows || !soft_scores) return IA_E_ARG;
    if (!(sigma != 0.0f)) return IA_E_ARG;                  // ZeroDivisionError in the reference
    SoftArgs a;
    a.boxes = boxes; a.scores_t = scores_t; a.keep_count = keep_count; a.keep_rows = keep_rows;
    a.soft_scores = soft_scores; a.score_thr = score_thr;
    a.sp.iou_thr = iou_thr; a.sp.sigma = sigma; a.sp.min_score = min_score; a.sp.method = method;
    a.R = R; a.Rs = Rs; a.C = C; a.cap = soft_cap(R);
    hipLaunchKernelGGL(k_soft_nms, dim3((unsigned)C, (unsigned)batch), dim3(kSoftThreads),
                       soft_lds_bytes(a.cap), s, a);
    return hip_status(hipGetLastError());
}

// ------------------------------------------------------------------ the standalone op
struct SoftDetsSrc {
    const float *d;                // (n,5)
    __device__ __forceinline__ bool pred(uint32_t) const { return true; }
    __device__ __forceinline__ float score(uint32_t r) const { return d[5 * (size_t)r + 4]; }
    __device__ __forceinline__ float4 box(uint32_t r) const
    {
        const float *q = d + 5 * (size_t)r;
        return make_float4(q[0], q[1], q[2], q[3]);
    }
};
struct SoftDetsOut {
    float *out_dets;
    int32_t *out_inds, *count;
    __device__ __forceinline__ void emit(uint32_t slot, uint32_t row, float s, const float4 &t)
    {
        float *o = out_dets + 5 * (size_t)slot;
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; o[4] = s;
        out_inds[slot] = (int32_t)row;
    }
    __device__ __forceinline__ void done(uint32_t n) { *count = (int32_t)n; }
};

struct SoftSingleArgs {
    const float *dets;
    float *out_dets;
    int32_t *out_inds, *count;
    SoftParams sp;
    int32_t n, cap;
};

__global__ void __launch_bounds__(kSoftThreads) k_soft_nms_single(SoftSingleArgs a)
{
    extern __shared__ float s_dyn[];
    __shared__ SoftSmem sm;
    float *s_score = s_dyn;
    uint16_t *s_elem = reinterpret_cast<uint16_t *>(s_dyn + a.cap);
    uint16_t *s_list = s_elem + a.cap;
    SoftDetsSrc src{a.dets};
    SoftDetsOut out{a.out_dets, a.out_inds, a.count};
    soft_nms_block((uint32_t)a.n, src, out, a.sp, s_score, s_elem, s_list, sm);
}

int launch_soft_nms_single(const float *dets, int n, float iou_thr, int method, float sigma,
                           float min_score, float *out_dets, int32_t *out_inds, int32_t *count,
                           hipStream_t s)
{
    if (n > IA_MAX_CANDIDATES) return IA_E_LIMIT_BOXES;
    if (n < 0 || !count) return IA_E_ARG;
    if (!(sigma != 0.0f)) return IA_E_ARG;
    if (n == 0) {
        hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t), s);
        return hip_status(e);
    }
    if (!dets || !out_dets || !out_inds) return IA_E_ARG;
    SoftSingleArgs a;
    a.dets = dets; a.out_dets = out_dets; a.out_inds = out_inds; a.count = count;
    a.sp.iou_thr = iou_thr; a.sp.sigma = sigma; a.sp.min_score = min_score; a.sp.method = method;
    a.n = n; a.cap = soft_cap(n);
    hipLaunchKernelGGL(k_soft_nms_single, dim3(1), dim3(kSoftThreads), soft_lds_bytes(a.cap), s, a);
    return hip_status(hipGetLastError());
}

}  // namespace ia
