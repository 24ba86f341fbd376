// IoUawareRetinaHead.loss for ALL pyramid levels at once
// (reference iou_aware_retina_head.py:221-313 `loss_single` x 5 levels, :315-387 `loss`;
// core/loss/losses.py:226-247,279-303 focal, :385-411 smooth-L1, :460-480 IoU BCE).
//
// The per-level kernels of loss.hip cost one launch per (loss, level, direction): 30 launches,
// the four small levels running far below the HBM stream of P3, plus ~140 scalar torch kernels
// of autograd glue around them.  Here one training iteration's loss part is
//
//   forward : k_pack_targets  k_focal_ml<fwd>  k_box_ml<fwd>  k_headloss_finalize   (4 launches)
//   backward: k_focal_ml<bwd>  k_box_ml<bwd>                                        (2 launches)
//
// * k_focal_ml -- focal loss (gamma = 2) over the class logits of every level: the wavefront
//   tiling of k_rowmax / k_focal (wavefront = anchor x 256 positions, every class-plane access one
//   contiguous 1 KiB segment), blocks of the SMALL levels first so that their latency-bound
//   tails overlap the P3 stream.  All elements are evaluated as negatives (no per-element label
//   compare / select; the per-anchor weight is applied once per position after the class loop)
//   and the rare positive element of a positive anchor is corrected afterwards: five plain VALU
//   operations + exp + rcp + log per element, evaluated eight elements abreast so that the
//   transcendental unit is fed back to back, class planes double-buffered in registers.
//   MI355X, B = 4 (rocprofv3): forward 52.5 us = 5.1 TB/s, backward 105 us = 5.0 TB/s of
//   algorithmic bytes (0.64 / 0.63 of the 8 TB/s peak); what it took, in measured steps
//   (forward): one wavefront per (anchor, tile) running all 80 classes on every level 99.5 us ->
//   class range of the small levels split 71 -> counted-wait ping-pong pipeline 66 -> labels
//   from an anchor-major copy instead of 72-byte-stride gathers 55 -> split target 4096: 52.5.
// * k_box_ml  -- smooth-L1 + IoU target + IoU BCE of every level in one pass over the box
//   deltas; anchors whose bbox_weights are zero (all negatives: > 99 %) contribute exactly 0
//   and skip the exact-math decode; backward writes d(bbox_pred) = smooth-L1 part + the part
//   through the attached IoU target in one store, and d(iou_pred).
// * k_headloss_finalize -- the 64 fp64 partial slots per (loss, level) -> fp32 losses
//   (sum / avg_factor) * loss_weight, avg_factor = sum_b max(n_pos_b, 1) read from the
//   assignment kernel's counts: the normaliser never visits the host.
#include <string.h>
#include "ia_loss.hpp"

namespace ia {

struct HLLevels {
    int32_t L, B, A, C;
    int32_t H[IA_MAX_LEVELS], W[IA_MAX_LEVELS], stride[IA_MAX_LEVELS];
    int32_t blk_off[IA_MAX_LEVELS + 1];   // prefix over launch order o (level = L-1-o) of B*A*tiles
    // focal kernel: the class range of the small levels is cut into csplit chunks of cchunk classes
    // (a lone wavefront per (image, anchor, tile) would run 80 dependent class steps while the
    // big level streams: the small levels' chains, not HBM, would set the kernel's duration)
    int32_t csplit[IA_MAX_LEVELS], cchunk[IA_MAX_LEVELS];
    int32_t fblk_off[IA_MAX_LEVELS + 1];  // prefix of B*A*csplit*tiles
    int32_t pack_off[IA_MAX_LEVELS + 1];  // prefix of B*A*HW: element offset of a level in the packed targets
};

struct BlockRef { int l, b, an, p0, chunk; };

template <bool SPLIT>
__device__ __forceinline__ BlockRef locate_block(const HLLevels &lv, int bid)
{
    const int32_t *off = SPLIT ? lv.fblk_off : lv.blk_off;
    int o = 0;
    while (bid >= off[o + 1]) ++o;
    BlockRef r;
    r.l = lv.L - 1 - o;
    int q = bid - off[o];
    const int tiles = (lv.H[r.l] * lv.W[r.l] + 255) / 256;
    const int tile = q % tiles; q /= tiles;
    r.chunk = 0;
    r.p0 = tile * 256;
    if (SPLIT) { r.chunk = q % lv.csplit[r.l]; q /= lv.csplit[r.l]; }
    r.an = q % lv.A;
    r.b = q / lv.A;
    return r;
}

constexpr int kSlotMask = IA_LOSS_SLOTS - 1;
constexpr int64_t kFocalLevelWaves = 4096;
constexpr int kNumLoss = 3;               // cls, bbox, iou
// layout of the fp32 result / upstream-gradient vector: [k * L + l] per-level, [3L + k] totals,
// [3L + 3] avg_factor
__device__ __forceinline__ float upstream(const float *gin, const float *res, int L, int k, int l,
                                          float lw)
{
    return ((gin[k * L + l] + gin[3 * L + k]) * lw) / res[3 * L + 3];
}

// ------------------------------------------------------------------ focal, all levels
struct FocalMLArgs {
    HLLevels lv;
    const void *cls[IA_MAX_LEVELS];
    const int32_t *lab_am;                // packed targets: anchor-major labels / weights,
    const float *w_am;                    // level l at pack_off[l], then (B, A, HW)
    float *grad[IA_MAX_LEVELS];
    double *sums;                         // fwd: [3][L][IA_LOSS_SLOTS]
    const float *gin, *res;               // bwd
    float alpha_pos, alpha_neg, loss_weight;
    int32_t big_logits;                   // evaluate the exact tail for logits > kXMax (fwd)
};

// ---- element math.  With t = exp(x), s = 1 + t:   sigmoid(x) = t/s,  1 - sigmoid(x) = 1/s,
// BCE(x, 0) = softplus(x) = log(s),  BCE(x, 1) = softplus(-x) = log(s) - x.
// One v_exp_f32, one v_rcp_f32, one v_log_f32 per element (the transcendental unit issues at an
// eighth of the plain VALU rate on gfx950: these three are 48 of the ~60 issue cycles of an
// element) and five plain operations; the factor ln 2 of log2 -> log and the anchor's weight are
// applied once per position after the class loop.  x is clamped to kXMax before the exponential
// (exp(88.8) overflows fp32); above it sigmoid = 1 and softplus(x) = x hold to the last bit, and
// the (never observed) logits beyond it take the exact branch below.
constexpr float kXMax = 60.0f;
constexpr float kLog2e = 1.44269504088896341f, kLn2 = 0.693147180559945309f;

struct Sig { float p, q, lg; };          // sigmoid, 1 - sigmoid, log2(1 + exp(x))
__device__ __forceinline__ Sig sig_parts(float x)
{
    const float t = __builtin_amdgcn_exp2f(__builtin_fminf(x, kXMax) * kLog2e);
    const float s = 1.0f + t;
    Sig r;
    r.q = __builtin_amdgcn_rcpf(s);
    r.lg = __builtin_amdgcn_logf(s);
    r.p = t * r.q;
    return r;
}
// UNWEIGHTED negative element  p^2 * BCE(x, 0), in units of ln 2, and its x-derivative (natural units)
__device__ __forceinline__ float neg_val2(const Sig &g) { return (g.p * g.p) * g.lg; }
__device__ __forceinline__ float neg_der(const Sig &g)
{
    return (g.p * g.p) * __builtin_fmaf(2.0f * kLn2, g.lg * g.q, g.p);
}
// positive element  q^2 * BCE(x, 1) (natural units) and its derivative
__device__ __forceinline__ float pos_val(const Sig &g, float x)
{
    return (g.q * g.q) * __builtin_fmaf(kLn2, g.lg, -x);
}
__device__ __forceinline__ float pos_der(const Sig &g, float x)
{
    return -((g.q * g.q) * __builtin_fmaf(2.0f * __builtin_fmaf(kLn2, g.lg, -x), g.p, g.q));
}

template <typename T> struct MLPack;
template <> struct MLPack<float> {
    typedef float V __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void unpack(const V &q, float (&v)[4]) { v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
};
template <> struct MLPack<uint16_t> {
    typedef uint16_t V __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void unpack(const V &q, float (&v)[4])
    {
        v[0] = bf16_to_f32(q.x); v[1] = bf16_to_f32(q.y); v[2] = bf16_to_f32(q.z); v[3] = bf16_to_f32(q.w);
    }
};

template <typename T, bool BWD>
__global__ void __launch_bounds__(64) k_focal_ml(FocalMLArgs a)
{
    const int lane = threadIdx.x;
    const BlockRef r = locate_block<true>(a.lv, blockIdx.x);
    const int A = a.lv.A, C = a.lv.C, HW = a.lv.H[r.l] * a.lv.W[r.l];
    const int cbeg = r.chunk * a.lv.cchunk[r.l];
    const int cend = (cbeg + a.lv.cchunk[r.l] < C) ? (cbeg + a.lv.cchunk[r.l]) : C;
    const T *cls = static_cast<const T *>(a.cls[r.l]) + ((size_t)r.b * A + r.an) * C * HW;
    float *grad = BWD ? a.grad[r.l] + ((size_t)r.b * A + r.an) * C * HW : nullptr;
    const float gs = BWD ? upstream(a.gin, a.res, a.lv.L, 0, r.l, a.loss_weight) : 1.0f;
    const bool vec = (HW & 3) == 0;
    // labels / weights of this anchor's 256 positions from the anchor-major packed copy
    // (k_pack_targets): two coalesced 16-byte loads per lane instead of eight 64-line gathers
    const size_t am = (size_t)a.lv.pack_off[r.l] + ((size_t)r.b * A + r.an) * HW;
    int pos[4], pc[4], lab[4];
    float wn[4], wp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pos[j] = vec ? (r.p0 + lane * 4 + j) : (r.p0 + lane + 64 * j);
        pc[j] = (pos[j] < HW) ? pos[j] : (vec ? (HW - 4 + j) : (HW - 1));   // clamped: loads unconditional
    }
    if (vec) {
        const int4 l4 = *reinterpret_cast<const int4 *>(a.lab_am + am + pc[0]);
        const float4 w4 = *reinterpret_cast<const float4 *>(a.w_am + am + pc[0]);
        lab[0] = l4.x; lab[1] = l4.y; lab[2] = l4.z; lab[3] = l4.w;
        wn[0] = w4.x; wn[1] = w4.y; wn[2] = w4.z; wn[3] = w4.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { lab[j] = a.lab_am[am + pc[j]]; wn[j] = a.w_am[am + pc[j]]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float w0 = (pos[j] < HW) ? wn[j] : 0.0f;                      // padding lanes weigh 0
        if (pos[j] >= HW) lab[j] = 0;
        wn[j] = (a.alpha_neg * w0) * gs;
        wp[j] = (a.alpha_pos * w0) * gs;
    }
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int K = 8;                                   // class planes in flight per wavefront
    if (vec) {
        // Software pipeline over groups of K class planes with two register buffers in ping-pong
        // (no register copies, so the waits stay counted): while one group is evaluated the next
        // group's 8 KiB per wavefront are in flight.
        using V = typename MLPack<T>::V;
        const T *src = cls + pc[0];
        auto issue = [&](V (&q)[K], int c0) {
#pragma unroll
            for (int i = 0; i < K; ++i) {                  // unconditional, clamped: no branches
                const int c = (c0 + i < cend) ? (c0 + i) : (cend - 1);
                q[i] = __builtin_nontemporal_load(reinterpret_cast<const V *>(src + (size_t)c * HW));
            }
        };
        // two class planes (8 elements per lane) at a time, stage by stage: the eight
        // exponentials, then the eight reciprocals and logarithms, issue back to back instead
        // of waiting on one element's dependent chain
        auto eval = [&](const V (&q)[K], int c0) {
#pragma unroll
            for (int i = 0; i < K; i += 2) {
                if (c0 + i < cend) {                       // wave-uniform
                    float v[8], t[8], rq[8], lg[8];
                    MLPack<T>::unpack(q[i], *reinterpret_cast<float (*)[4]>(&v[0]));
                    MLPack<T>::unpack(q[i + 1], *reinterpret_cast<float (*)[4]>(&v[4]));
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        t[k] = __builtin_amdgcn_exp2f(__builtin_fminf(v[k], kXMax) * kLog2e);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float sk = 1.0f + t[k];
                        rq[k] = __builtin_amdgcn_rcpf(sk);
                        lg[k] = __builtin_amdgcn_logf(sk);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const bool second = c0 + i + 1 < cend;
                    float o[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        Sig g; g.q = rq[k]; g.lg = lg[k]; g.p = t[k] * rq[k];
                        if (BWD) o[k] = neg_der(g) * wn[k & 3];
                        else if (k < 4 || second) acc[k & 3] += neg_val2(g);
                    }
                    if (BWD && pos[0] < HW) {
                        typedef float F4 __attribute__((ext_vector_type(4)));
                        F4 g0, g1;
                        g0.x = o[0]; g0.y = o[1]; g0.z = o[2]; g0.w = o[3];
                        g1.x = o[4]; g1.y = o[5]; g1.z = o[6]; g1.w = o[7];
                        F4 *d0 = reinterpret_cast<F4 *>(grad + (size_t)(c0 + i) * HW + pc[0]);
                        F4 *d1 = reinterpret_cast<F4 *>(grad + (size_t)(c0 + i + 1) * HW + pc[0]);
                        *d0 = g0;              // (non-temporal stores: no difference, 104.9 vs 105.5 us)
                        if (second) *d1 = g1;
                    }
                }
            }
        };
        V qa[K], qb[K];
        issue(qa, cbeg);
        for (int c0 = cbeg; c0 < cend; c0 += 2 * K) {
            issue(qb, c0 + K);
            eval(qa, c0);
            issue(qa, c0 + 2 * K);
            eval(qb, c0 + K);
        }
    } else {                                               // plane bases only 4-byte aligned
        for (int c0 = cbeg; c0 < cend; c0 += K) {
            float v[K][4];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int c = (c0 + i < cend) ? (c0 + i) : (cend - 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] = load_f32<T>(cls + (size_t)c * HW + pc[j]);
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (c0 + i < cend) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const Sig g = sig_parts(v[i][j]);
                        if (BWD) {
                            if (pos[j] < HW) grad[(size_t)(c0 + i) * HW + pos[j]] = neg_der(g) * wn[j];
                        } else acc[j] += neg_val2(g);
                    }
                }
            }
        }
    }
    // Corrections, all rare: (1) the positive element of a positive anchor replaces its
    // negative-form contribution (the same lane wrote the negative-form gradient above, so the
    // overwrite is ordered); (2) logits above kXMax: softplus(x) = x there, the clamp gave kXMax.
    float total = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float fix = 0.0f;
        if (lab[j] > cbeg && lab[j] <= cend && pos[j] < HW) {      // class lab-1 in [cbeg, cend)
            const size_t e = (size_t)(lab[j] - 1) * HW + pos[j];
            const float x = load_f32<T>(cls + e);
            const Sig g = sig_parts(x);
            if (BWD) grad[e] = pos_der(g, __builtin_fminf(x, kXMax)) * wp[j];
            else { acc[j] -= neg_val2(g); fix = pos_val(g, __builtin_fminf(x, kXMax)) * wp[j]; }
        }
        if (!BWD) total += __builtin_fmaf(acc[j] * kLn2, wn[j], fix);
    }
    if (!BWD && __builtin_expect(a.big_logits != 0, 0)) {
        // exact tail for logits > kXMax (requested by the host when it cannot exclude them):
        // add (x - kXMax) per such negative element
        for (int c = cbeg; c < cend; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (pos[j] < HW) {
                    const float x = load_f32<T>(cls + (size_t)c * HW + pos[j]);
                    if (x > kXMax && lab[j] != c + 1) total += (x - kXMax) * wn[j];
                }
    }
    if (!BWD) {
        const double d = wave_sum((double)total);
        if (lane == 0) atomicAdd(a.sums + (size_t)(0 * a.lv.L + r.l) * IA_LOSS_SLOTS +
                                     (blockIdx.x & kSlotMask), d);
    }
}

// ------------------------------------------------------------------ targets -> anchor-major
// labels / label_weights arrive position-major (n = p*A + a, the reference's order): read per
// anchor they are 8-byte gathers at a 72-byte stride, 64 cache lines per load instruction -- as
// many L1 line requests as the class stream itself.  One small pass transposes them (through LDS,
// both sides coalesced) to (B, A, HW) int32 / fp32; forward and backward read that copy.
struct PackArgs {
    HLLevels lv;
    const int64_t *labels[IA_MAX_LEVELS];
    const float *lw[IA_MAX_LEVELS];
    int32_t *lab_am;
    float *w_am;
    int32_t tile_off[IA_MAX_LEVELS + 1];  // prefix of B * tiles_l
};

__global__ void __launch_bounds__(256) k_pack_targets(PackArgs a)
{
    __shared__ int32_t s_lab[IA_MAX_ANCHORS * 257];
    __shared__ float s_w[IA_MAX_ANCHORS * 257];
    int l = 0;
    while ((int)blockIdx.x >= a.tile_off[l + 1]) ++l;
    int q = blockIdx.x - a.tile_off[l];
    const int A = a.lv.A, HW = a.lv.H[l] * a.lv.W[l];
    const int tiles = (HW + 255) / 256;
    const int b = q / tiles, p0 = (q - b * tiles) * 256;
    const int npos = (HW - p0 < 256) ? (HW - p0) : 256;
    const size_t base = ((size_t)b * HW + p0) * A;
    for (int k = threadIdx.x; k < npos * A; k += 256) {
        const int p = k / A, an = k - p * A;
        s_lab[an * 257 + p] = (int32_t)a.labels[l][base + k];
        s_w[an * 257 + p] = a.lw[l][base + k];
    }
    __syncthreads();
    const size_t out = (size_t)a.lv.pack_off[l] + (size_t)b * A * HW + p0;
    for (int k = threadIdx.x; k < 256 * A; k += 256) {
        const int an = k >> 8, p = k & 255;
        if (p < npos) {
            a.lab_am[out + (size_t)an * HW + p] = s_lab[an * 257 + p];
            a.w_am[out + (size_t)an * HW + p] = s_w[an * 257 + p];
        }
    }
}

// ------------------------------------------------------------------ smooth-L1 + IoU BCE, all levels
struct BoxMLArgs {
    HLLevels lv;
    BaseAnchors ba;
    const void *reg[IA_MAX_LEVELS], *iou[IA_MAX_LEVELS];
    const float *bt[IA_MAX_LEVELS], *bw[IA_MAX_LEVELS];
    float *g_reg[IA_MAX_LEVELS], *g_iou[IA_MAX_LEVELS];
    double *sums;
    const float *gin, *res;
    float means[4], stds[4];
    float beta, lw_bbox, lw_iou;
    int32_t attach;
};

template <typename T, bool BWD>
__global__ void __launch_bounds__(256) k_box_ml(BoxMLArgs a)
{
    __shared__ double red[2][4];
    const BlockRef r = locate_block<false>(a.lv, blockIdx.x);
    const int A = a.lv.A, W = a.lv.W[r.l], HW = a.lv.H[r.l] * W;
    const int p = r.p0 + threadIdx.x;
    double acc_l1 = 0.0, acc_iou = 0.0;
    if (p < HW) {
        const size_t ba = (size_t)r.b * A + r.an;
        const size_t e = ba * HW + p;
        const size_t n = ((size_t)r.b * HW + p) * A + r.an;
        const float4 wt4 = reinterpret_cast<const float4 *>(a.bw[r.l])[n];
        const float wv[4] = {wt4.x, wt4.y, wt4.z, wt4.w};
        const bool live = (wv[0] != 0.0f) | (wv[1] != 0.0f) | (wv[2] != 0.0f) | (wv[3] != 0.0f);
        float g_box[4] = {0.0f, 0.0f, 0.0f, 0.0f}, g_iou = 0.0f;
        if (live) {
            const T *bp = static_cast<const T *>(a.reg[r.l]) + ba * 4 * HW + p;
            const float dp[4] = {load_f32<T>(bp), load_f32<T>(bp + (size_t)HW),
                                 load_f32<T>(bp + (size_t)2 * HW), load_f32<T>(bp + (size_t)3 * HW)};
            const float4 tq = reinterpret_cast<const float4 *>(a.bt[r.l])[n];
            const float dt[4] = {tq.x, tq.y, tq.z, tq.w};
            const int y = p / W, x = p - y * W;
            const float sx = (float)(x * a.lv.stride[r.l]), sy = (float)(y * a.lv.stride[r.l]);
            const float *b4 = a.ba.v[r.l][r.an];
            const float anc[4] = {b4[0] + sx, b4[1] + sy, b4[2] + sx, b4[3] + sy};
            const IouElem q = iou_target_elem(anc, dp, dt, a.means, a.stds);
            const float xl = load_f32<T>(static_cast<const T *>(a.iou[r.l]) + e);
            if (!BWD) {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) s += smooth_l1_val(dp[k] - dt[k], a.beta) * wv[k];
                acc_l1 = (double)s;
                acc_iou = (double)(bce_logits_(xl, q.t) * wv[0]);
            } else {
                const float gs1 = upstream(a.gin, a.res, a.lv.L, 1, r.l, a.lw_bbox);
                const float gs2 = upstream(a.gin, a.res, a.lv.L, 2, r.l, a.lw_iou);
                g_iou = ((sigmoidf_(xl) - q.t) * wv[0]) * gs2;
                float gv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (a.attach) iou_bce_box_grad(q, xl, wv[0], gs2, a.stds, gv);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    g_box[k] = (smooth_l1_der(dp[k] - dt[k], a.beta) * wv[k]) * gs1 + gv[k];
            }
        }
        if (BWD) {
            float *go = a.g_reg[r.l] + ba * 4 * HW + p;
            go[0] = g_box[0];
            go[(size_t)HW] = g_box[1];
            go[(size_t)2 * HW] = g_box[2];
            go[(size_t)3 * HW] = g_box[3];
            a.g_iou[r.l][e] = g_iou;
        }
    }
    if (!BWD) {
        // positives are rare: most workgroups have nothing to add
        const bool any = __syncthreads_or((acc_l1 != 0.0) | (acc_iou != 0.0));
        if (any) {
            const double s1 = wave_sum(acc_l1), s2 = wave_sum(acc_iou);
            const int w = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) { red[0][w] = s1; red[1][w] = s2; }
            __syncthreads();
            if (threadIdx.x < 2) {
                const double s = ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) +
                                 red[threadIdx.x][3];
                if (s != 0.0)
                    atomicAdd(a.sums + (size_t)((1 + threadIdx.x) * a.lv.L + r.l) * IA_LOSS_SLOTS +
                                  (blockIdx.x & kSlotMask), s);
            }
        }
    }
}

// ------------------------------------------------------------------ channels-last head outputs
// The training head (winograd_train.py) produces channels-last tensors: element (b, p, a, c) of a
// class map sits at (b*HW + p) * pix_stride + a*C + c, i.e. in the reference's flattened order
// (cls_score.permute(0, 2, 3, 1).reshape(-1, C), iou_aware_retina_head.py:236-240) -- the targets
// (anchor-major n = (b, p, a)) index it directly, no packed copy, and every load / store is a
// 16-byte piece of a contiguous run.  reg / iou may be channel slices of one wider tensor
// (pix_stride > A*4 / A).  fp32 only.
struct NhwcLevels {
    int32_t L, B, A, C;
    int32_t H[IA_MAX_LEVELS], W[IA_MAX_LEVELS], stride[IA_MAX_LEVELS];
    int32_t fblk_off[IA_MAX_LEVELS + 1];      // focal: blocks of kFocalChunks float4 chunks, launch order
    int32_t bblk_off[IA_MAX_LEVELS + 1];      // box: blocks of 256 anchors, launch order
};
constexpr int kFocalU = 4;                    // float4 chunks per thread
constexpr int kFocalChunks = 256 * kFocalU;   // per block

struct FocalNhwcArgs {
    NhwcLevels lv;
    const float *cls[IA_MAX_LEVELS];
    int64_t ps_cls[IA_MAX_LEVELS], ps_grad[IA_MAX_LEVELS];   // pixel strides (elements)
    const int64_t *labels[IA_MAX_LEVELS];
    const float *lw[IA_MAX_LEVELS];
    float *grad[IA_MAX_LEVELS];
    double *sums;
    const float *gin, *res;
    float alpha_pos, alpha_neg, loss_weight;
    int32_t big_logits;
};

template <bool BWD>
__global__ void __launch_bounds__(256) k_focal_nhwc(FocalNhwcArgs a)
{
    __shared__ double red[4];
    int o = 0;
    while ((int)blockIdx.x >= a.lv.fblk_off[o + 1]) ++o;
    const int l = a.lv.L - 1 - o;
    const int A = a.lv.A, C = a.lv.C, C4 = C >> 2, AC4 = A * C4;
    const int64_t HW = (int64_t)a.lv.H[l] * a.lv.W[l];
    const int64_t nchunks = (int64_t)a.lv.B * HW * AC4;
    // chunk -> (pixel, anchor, class quad) without per-chunk integer divisions (a 64-bit division
    // per chunk cost as many issue slots as the loss math of its four elements): one division
    // per workgroup for its first chunk, then offsets < AC4 + 1024 divided through the float
    // reciprocal ((n + 0.5) / d is at least 0.5 / d away from an integer, far above fp32
    // rounding for n, d < 2^14)
    const int64_t base0 = (int64_t)(blockIdx.x - a.lv.fblk_off[o]) * kFocalChunks;
    const int64_t pix0 = base0 / AC4;
    const int r0 = (int)(base0 - pix0 * AC4);
    const float inv_ac4 = 1.0f / (float)AC4, inv_c4 = 1.0f / (float)C4;
    const float gs = BWD ? upstream(a.gin, a.res, a.lv.L, 0, l, a.loss_weight) : 1.0f;
    const float *cls = a.cls[l];
    const int64_t ps = a.ps_cls[l], pg = BWD ? a.ps_grad[l] : 0;
    // every load of the thread is issued before the first use (addresses clamped, no predicate):
    // a load next to its use, or under `on ? load : 0`, compiles to load + s_waitcnt vmcnt(0)
    // per piece, i.e. eight dependent memory round trips per thread instead of one
    float4 v[kFocalU];
    int32_t labv[kFocalU];
    float lwv[kFocalU];
    int cq[kFocalU];
    int64_t goff[kFocalU];
    bool on[kFocalU];
    const int32_t *labels = reinterpret_cast<const int32_t *>(a.labels[l]);   // low words: labels < 2^31
    const float *lwp = a.lw[l];
#pragma unroll
    for (int u = 0; u < kFocalU; ++u) {
        int j = (int)threadIdx.x + 256 * u;
        on[u] = base0 + j < nchunks;
        if (!on[u]) j = (int)(nchunks - 1 - base0);         // clamped: loads unconditional
        const int rr = r0 + j;
        const int dp = (int)(((float)rr + 0.5f) * inv_ac4);
        const int64_t pix = pix0 + dp;                      // (b, p)
        const int r = rr - dp * AC4;                        // a * C4 + class quad
        const int an = (int)(((float)r + 0.5f) * inv_c4);
        const int64_t anchor = pix * A + an;
        cq[u] = r - an * C4;
        typedef float F4 __attribute__((ext_vector_type(4)));
        const F4 q = __builtin_nontemporal_load(reinterpret_cast<const F4 *>(cls + pix * ps + 4 * r));
        v[u] = make_float4(q.x, q.y, q.z, q.w);
        labv[u] = labels[2 * anchor];
        lwv[u] = lwp[anchor];
        goff[u] = pix * pg + 4 * r;
    }
    double total = 0.0;
#pragma unroll
    for (int u = 0; u < kFocalU; ++u) {
        const int lab = labv[u];
        const float w0 = on[u] ? lwv[u] : 0.0f;
        const float wn = (a.alpha_neg * w0) * gs, wp = (a.alpha_pos * w0) * gs;
        const float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        float t[4], rq[4], lg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_exp2f(__builtin_fminf(x[k], kXMax) * kLog2e);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sk = 1.0f + t[k];
            rq[k] = __builtin_amdgcn_rcpf(sk);
            lg[k] = __builtin_amdgcn_logf(sk);
        }
        // every element in its negative form first (no per-element select: the forward kernel is
        // VALU-bound -- 3 transcendentals + the plain operations of an element are ~60 us of
        // issue time at batch 4, against 41 us of HBM time); the one positive element of a
        // positive anchor (0.1 % of the anchors) is corrected after, under a rare branch
        float o4[4], acc = 0.0f, fix = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            Sig g; g.q = rq[k]; g.lg = lg[k]; g.p = t[k] * rq[k];
            if (BWD) o4[k] = neg_der(g) * wn;
            else acc += neg_val2(g);
        }
        const int jp = lab - 1 - 4 * cq[u];                 // the positive class's slot in this quad
        if (jp >= 0 && jp < 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k == jp) {
                    Sig g; g.q = rq[k]; g.lg = lg[k]; g.p = t[k] * rq[k];
                    if (BWD) o4[k] = pos_der(g, __builtin_fminf(x[k], kXMax)) * wp;
                    else { acc -= neg_val2(g); fix = pos_val(g, __builtin_fminf(x[k], kXMax)) * wp; }
                }
        }
        if (!BWD && a.big_logits) {                         // exact tail, on request (wave-uniform)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x[k] > kXMax && k != jp) fix += (x[k] - kXMax) * wn;
        }
        if (BWD) {
            if (on[u]) *reinterpret_cast<float4 *>(a.grad[l] + goff[u]) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        } else total += (double)__builtin_fmaf(acc * kLn2, wn, fix);
    }
    if (!BWD) {
        const double d = wave_sum(total);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicAdd(a.sums + (size_t)(0 * a.lv.L + l) * IA_LOSS_SLOTS + (blockIdx.x & kSlotMask),
                      (red[0] + red[1]) + (red[2] + red[3]));
    }
}

struct BoxNhwcArgs {
    NhwcLevels lv;
    BaseAnchors ba;
    const float *reg[IA_MAX_LEVELS], *iou[IA_MAX_LEVELS];
    int64_t ps_reg[IA_MAX_LEVELS], ps_iou[IA_MAX_LEVELS], pg_reg[IA_MAX_LEVELS], pg_iou[IA_MAX_LEVELS];
    const float *bt[IA_MAX_LEVELS], *bw[IA_MAX_LEVELS];
    float *g_reg[IA_MAX_LEVELS], *g_iou[IA_MAX_LEVELS];
    double *sums;
    const float *gin, *res;
    float means[4], stds[4];
    float beta, lw_bbox, lw_iou;
    int32_t attach;
    int32_t g_pad[IA_MAX_LEVELS];         // bwd: zero-gradient channels behind d(iou) in the same pixel row
};

template <bool BWD>
__global__ void __launch_bounds__(256) k_box_nhwc(BoxNhwcArgs a)
{
    __shared__ double red[2][4];
    int o = 0;
    while ((int)blockIdx.x >= a.lv.bblk_off[o + 1]) ++o;
    const int l = a.lv.L - 1 - o;
    const int A = a.lv.A, W = a.lv.W[l];
    const int64_t HW = (int64_t)a.lv.H[l] * W, N = (int64_t)a.lv.B * HW * A;
    const int64_t n = (int64_t)(blockIdx.x - a.lv.bblk_off[o]) * 256 + threadIdx.x;   // (b, p, a)
    double acc_l1 = 0.0, acc_iou = 0.0;
    if (n < N) {
        const int64_t pix = n / A;
        const int an = (int)(n - pix * A);
        const int p = (int)(pix % HW);
        const float4 wt4 = reinterpret_cast<const float4 *>(a.bw[l])[n];
        const float wv[4] = {wt4.x, wt4.y, wt4.z, wt4.w};
        const bool live = (wv[0] != 0.0f) | (wv[1] != 0.0f) | (wv[2] != 0.0f) | (wv[3] != 0.0f);
        float g_box[4] = {0.0f, 0.0f, 0.0f, 0.0f}, g_iou = 0.0f;
        if (live) {
            const float4 d4 = *reinterpret_cast<const float4 *>(a.reg[l] + pix * a.ps_reg[l] + 4 * an);
            const float dp[4] = {d4.x, d4.y, d4.z, d4.w};
            const float4 tq = reinterpret_cast<const float4 *>(a.bt[l])[n];
            const float dt[4] = {tq.x, tq.y, tq.z, tq.w};
            const int y = p / W, x = p - y * W;
            const float sx = (float)(x * a.lv.stride[l]), sy = (float)(y * a.lv.stride[l]);
            const float *b4 = a.ba.v[l][an];
            const float anc[4] = {b4[0] + sx, b4[1] + sy, b4[2] + sx, b4[3] + sy};
            const IouElem q = iou_target_elem(anc, dp, dt, a.means, a.stds);
            const float xl = a.iou[l][pix * a.ps_iou[l] + an];
            if (!BWD) {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) s += smooth_l1_val(dp[k] - dt[k], a.beta) * wv[k];
                acc_l1 = (double)s;
                acc_iou = (double)(bce_logits_(xl, q.t) * wv[0]);
            } else {
                const float gs1 = upstream(a.gin, a.res, a.lv.L, 1, l, a.lw_bbox);
                const float gs2 = upstream(a.gin, a.res, a.lv.L, 2, l, a.lw_iou);
                g_iou = ((sigmoidf_(xl) - q.t) * wv[0]) * gs2;
                float gv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (a.attach) iou_bce_box_grad(q, xl, wv[0], gs2, a.stds, gv);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    g_box[k] = (smooth_l1_der(dp[k] - dt[k], a.beta) * wv[k]) * gs1 + gv[k];
            }
        }
        if (BWD) {
            *reinterpret_cast<float4 *>(a.g_reg[l] + pix * a.pg_reg[l] + 4 * an) =
                make_float4(g_box[0], g_box[1], g_box[2], g_box[3]);
            a.g_iou[l][pix * a.pg_iou[l] + an] = g_iou;
            // d(reg) | d(iou) as slices of one wider tensor: its alignment channels behind the IoU
            // slice get their zero gradient here (the last anchor's thread), not from a fill per level
            if (an == A - 1)
                for (int k = 0; k < a.g_pad[l]; ++k) a.g_iou[l][pix * a.pg_iou[l] + A + k] = 0.0f;
        }
    }
    if (!BWD) {
        const bool any = __syncthreads_or((acc_l1 != 0.0) | (acc_iou != 0.0));
        if (any) {
            const double s1 = wave_sum(acc_l1), s2 = wave_sum(acc_iou);
            const int w = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) { red[0][w] = s1; red[1][w] = s2; }
            __syncthreads();
            if (threadIdx.x < 2) {
                const double s = ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) +
                                 red[threadIdx.x][3];
                if (s != 0.0)
                    atomicAdd(a.sums + (size_t)((1 + threadIdx.x) * a.lv.L + l) * IA_LOSS_SLOTS +
                                  (blockIdx.x & kSlotMask), s);
            }
        }
    }
}

// ------------------------------------------------------------------ slots -> losses
struct FinArgs {
    const double *sums;
    const int32_t *counts;                // (B, 2) from ia_anchor_targets, or NULL
    const float *avg_dev;                 // device scalar, or NULL
    float avg_host;
    float lw[kNumLoss];
    int32_t L, B;
    float *res;                           // 3L + 4
};

__global__ void __launch_bounds__(64) k_headloss_finalize(FinArgs a)
{
    __shared__ float s_loss[kNumLoss * IA_MAX_LEVELS];
    float avg = a.avg_host;
    if (a.counts) {                       // sum_i max(n_pos_i, 1)   (anchor_target.py:94)
        int tot = 0;
        for (int b = 0; b < a.B; ++b) tot += (a.counts[2 * b] > 1) ? a.counts[2 * b] : 1;
        avg = (float)tot;
    } else if (a.avg_dev) avg = a.avg_dev[0];
    const int i = threadIdx.x;
    if (i < kNumLoss * a.L) {
        double s = 0.0;
        for (int k = 0; k < IA_LOSS_SLOTS; ++k) s += a.sums[(size_t)i * IA_LOSS_SLOTS + k];
        // weighted_*: sum()[None] / avg_factor, then * loss_weight (losses.py:303,411,480)
        const float v = a.lw[i / a.L] * ((float)s / avg);
        s_loss[i] = v;
        a.res[i] = v;
    }
    __syncthreads();
    if (i < kNumLoss) {                   // parse_losses: sum over the levels, in level order
        float t = 0.0f;
        for (int l = 0; l < a.L; ++l) t += s_loss[i * a.L + l];
        a.res[kNumLoss * a.L + i] = t;
    }
    if (i == 0) a.res[kNumLoss * a.L + kNumLoss] = avg;
}

static int fill_levels(const ia_head_geom *g, int B, HLLevels &lv)
{
    if (!g || B < 1) return IA_E_ARG;
    if (g->num_levels < 1 || g->num_levels > IA_MAX_LEVELS) return IA_E_ARG;
    if (g->num_anchors < 1 || g->num_anchors > IA_MAX_ANCHORS || g->num_classes < 1) return IA_E_ARG;
    if (g->layout != IA_LAYOUT_NCHW) return IA_E_ARG;            // training kernels: NCHW only
    if (g->cls_activation != IA_CLS_SIGMOID) return IA_E_ARG;    // sigmoid focal loss only
    lv.L = g->num_levels; lv.B = B; lv.A = g->num_anchors; lv.C = g->num_classes;
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        const bool on = l < lv.L;
        if (on && (g->H[l] < 1 || g->W[l] < 1)) return IA_E_ARG;
        lv.H[l] = on ? g->H[l] : 0; lv.W[l] = on ? g->W[l] : 0; lv.stride[l] = on ? g->stride[l] : 0;
    }
    int64_t off = 0, foff = 0, poff = 0;
    lv.blk_off[0] = lv.fblk_off[0] = 0;
    for (int l = 0; l <= IA_MAX_LEVELS; ++l) {
        lv.pack_off[l] = (int32_t)poff;
        if (l < lv.L) poff += (int64_t)B * lv.A * lv.H[l] * lv.W[l];
        if (poff > 2147483647LL) return IA_E_ARG;
    }
    for (int l = 0; l < IA_MAX_LEVELS; ++l) { lv.csplit[l] = 1; lv.cchunk[l] = lv.C; }
    for (int o = 0; o < IA_MAX_LEVELS; ++o) {
        if (o < lv.L) {
            const int l = lv.L - 1 - o;
            const int64_t blocks = (int64_t)B * lv.A * (((int64_t)lv.H[l] * lv.W[l] + 255) / 256);
            // >= ~4 wavefronts per SIMD per level, chunks of whole 8-class load groups (MI355X,
            // B = 4: target 1 / 1024 / 2048 / 4096 wavefronts -> fwd 51.9 / 53.3 / 55.6 / 52.5 us,
            // bwd 110.9 / 112.2 / 112.2 / 108.2 us)
            int split = (int)((kFocalLevelWaves + blocks - 1) / blocks);   // blocks = wavefronts of the level
            if (split > (lv.C + 7) / 8) split = (lv.C + 7) / 8;
            if (split < 1) split = 1;
            lv.cchunk[l] = ((lv.C + split - 1) / split + 7) / 8 * 8;
            lv.csplit[l] = (lv.C + lv.cchunk[l] - 1) / lv.cchunk[l];
            off += blocks;
            foff += blocks * lv.csplit[l];
            if (foff > 2147483647LL) return IA_E_ARG;
        }
        lv.blk_off[o + 1] = (int32_t)off;
        lv.fblk_off[o + 1] = (int32_t)foff;
    }
    return 0;
}

}  // namespace ia

extern "C" {

size_t ia_head_loss_workspace_bytes(const ia_head_geom *g, int batch)
{
    ia::HLLevels lv;
    if (ia::fill_levels(g, batch, lv)) return 0;
    // fp64 slots | packed labels (int32) | packed weights (fp32), 256-byte aligned pieces
    const size_t slots = ((sizeof(double) * ia::kNumLoss * lv.L * IA_LOSS_SLOTS + 255) / 256) * 256;
    const size_t pk = (((size_t)lv.pack_off[lv.L] * 4 + 255) / 256) * 256;
    return slots + 2 * pk;
}

namespace ia {
static void carve(const HLLevels &lv, void *workspace, double *&sums, int32_t *&lab_am, float *&w_am)
{
    const size_t slots = ((sizeof(double) * kNumLoss * lv.L * IA_LOSS_SLOTS + 255) / 256) * 256;
    const size_t pk = (((size_t)lv.pack_off[lv.L] * 4 + 255) / 256) * 256;
    char *w = static_cast<char *>(workspace);
    sums = reinterpret_cast<double *>(w);
    lab_am = reinterpret_cast<int32_t *>(w + slots);
    w_am = reinterpret_cast<float *>(w + slots + pk);
}
}  // namespace ia

int ia_head_loss_fwd(const ia_head_geom *g, const ia_level_ptrs *p, int dtype, int batch,
                     const ia_head_targets *t, const ia_head_loss_cfg *cfg, void *workspace,
                     size_t workspace_bytes, float *result, void *stream)
{
    using namespace ia;
    if (!p || !t || !cfg || !workspace || !result || ((uintptr_t)workspace & 255u)) return IA_E_ARG;
    {
        const size_t need = ia_head_loss_workspace_bytes(g, batch);
        if (!need) return IA_E_ARG;
        if (workspace_bytes < need) return IA_E_WORKSPACE;
    }
    double *sums;
    FocalMLArgs fa;
    int rc = fill_levels(g, batch, fa.lv);
    if (rc) return rc;
    if (cfg->gamma != 2.0f || !(cfg->beta > 0.0f)) return IA_E_ARG;      // other gammas: per-level path
    const int L = fa.lv.L;
    BoxMLArgs ba;
    ba.lv = fa.lv;
    memcpy(ba.ba.v, g->base_anchors, sizeof(ba.ba.v));
    PackArgs pa;
    pa.lv = fa.lv;
    int32_t *lab_am; float *w_am;
    carve(fa.lv, workspace, sums, lab_am, w_am);
    pa.lab_am = lab_am; pa.w_am = w_am;
    fa.lab_am = lab_am; fa.w_am = w_am;
    pa.tile_off[0] = 0;
    for (int l = 0; l < IA_MAX_LEVELS; ++l)
        pa.tile_off[l + 1] = pa.tile_off[l] +
            (l < L ? batch * ((fa.lv.H[l] * fa.lv.W[l] + 255) / 256) : 0);
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        const bool on = l < L;
        if (on && (!p->cls[l] || !p->reg[l] || !p->iou[l] || !t->labels[l] || !t->label_weights[l] ||
                   !t->bbox_targets[l] || !t->bbox_weights[l]))
            return IA_E_ARG;
        fa.cls[l] = on ? p->cls[l] : nullptr;
        pa.labels[l] = on ? t->labels[l] : nullptr;
        pa.lw[l] = on ? t->label_weights[l] : nullptr;
        fa.grad[l] = nullptr;
        ba.reg[l] = on ? p->reg[l] : nullptr; ba.iou[l] = on ? p->iou[l] : nullptr;
        ba.bt[l] = on ? t->bbox_targets[l] : nullptr; ba.bw[l] = on ? t->bbox_weights[l] : nullptr;
        ba.g_reg[l] = ba.g_iou[l] = nullptr;
    }
    fa.sums = sums; fa.gin = fa.res = nullptr;
    fa.big_logits = cfg->exact_large_logits ? 1 : 0;
    fa.alpha_pos = cfg->alpha;
    fa.alpha_neg = (float)(1.0 - (double)cfg->alpha);   // python: (1 - alpha) in double, then fp32
    fa.loss_weight = cfg->loss_weight_cls;
    ba.sums = sums; ba.gin = ba.res = nullptr;
    for (int k = 0; k < 4; ++k) { ba.means[k] = g->means[k]; ba.stds[k] = g->stds[k]; }
    ba.beta = cfg->beta; ba.lw_bbox = cfg->loss_weight_bbox; ba.lw_iou = 1.0f;
    ba.attach = cfg->attach_iou_target ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * kNumLoss * (size_t)L * IA_LOSS_SLOTS, s);
    if (e != hipSuccess) return (int)e;
    const unsigned grid = (unsigned)fa.lv.blk_off[L], fgrid = (unsigned)fa.lv.fblk_off[L];
    hipLaunchKernelGGL(k_pack_targets, dim3((unsigned)pa.tile_off[L]), dim3(256), 0, s, pa);
    if (dtype == IA_F32) {
        hipLaunchKernelGGL((k_focal_ml<float, false>), dim3(fgrid), dim3(64), 0, s, fa);
        hipLaunchKernelGGL((k_box_ml<float, false>), dim3(grid), dim3(256), 0, s, ba);
    } else if (dtype == IA_BF16) {
        hipLaunchKernelGGL((k_focal_ml<uint16_t, false>), dim3(fgrid), dim3(64), 0, s, fa);
        hipLaunchKernelGGL((k_box_ml<uint16_t, false>), dim3(grid), dim3(256), 0, s, ba);
    } else return IA_E_ARG;
    FinArgs f;
    f.sums = sums; f.counts = t->counts; f.avg_dev = t->avg_factor_dev; f.avg_host = t->avg_factor;
    if (!f.counts && !f.avg_dev && !(f.avg_host > 0.0f)) return IA_E_ARG;
    f.lw[0] = cfg->loss_weight_cls; f.lw[1] = cfg->loss_weight_bbox; f.lw[2] = 1.0f;
    f.L = L; f.B = batch; f.res = result;
    hipLaunchKernelGGL(k_headloss_finalize, dim3(1), dim3(64), 0, s, f);
    return hip_status(hipGetLastError());
}

int ia_head_loss_bwd(const ia_head_geom *g, const ia_level_ptrs *p, int dtype, int batch,
                     const ia_head_targets *t, const ia_head_loss_cfg *cfg, const void *workspace,
                     const float *result, const float *grad_result, const ia_level_ptrs *grads,
                     void *stream)
{
    using namespace ia;
    if (!p || !t || !cfg || !workspace || !result || !grad_result || !grads) return IA_E_ARG;
    FocalMLArgs fa;
    int rc = fill_levels(g, batch, fa.lv);
    if (rc) return rc;
    if (cfg->gamma != 2.0f || !(cfg->beta > 0.0f)) return IA_E_ARG;
    const int L = fa.lv.L;
    BoxMLArgs ba;
    ba.lv = fa.lv;
    memcpy(ba.ba.v, g->base_anchors, sizeof(ba.ba.v));
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        const bool on = l < L;
        if (on && (!p->cls[l] || !p->reg[l] || !p->iou[l] || !t->labels[l] || !t->label_weights[l] ||
                   !t->bbox_targets[l] || !t->bbox_weights[l] || !grads->cls[l] || !grads->reg[l] ||
                   !grads->iou[l]))
            return IA_E_ARG;
        fa.cls[l] = on ? p->cls[l] : nullptr;
        fa.grad[l] = on ? (float *)grads->cls[l] : nullptr;
        ba.reg[l] = on ? p->reg[l] : nullptr; ba.iou[l] = on ? p->iou[l] : nullptr;
        ba.bt[l] = on ? t->bbox_targets[l] : nullptr; ba.bw[l] = on ? t->bbox_weights[l] : nullptr;
        ba.g_reg[l] = on ? (float *)grads->reg[l] : nullptr;
        ba.g_iou[l] = on ? (float *)grads->iou[l] : nullptr;
    }
    {
        double *sums_unused; int32_t *lab_am; float *w_am;
        carve(fa.lv, const_cast<void *>(workspace), sums_unused, lab_am, w_am);
        fa.lab_am = lab_am; fa.w_am = w_am;               // the forward call's packed targets
    }
    fa.sums = nullptr; fa.gin = grad_result; fa.res = result;
    fa.big_logits = 0;
    fa.alpha_pos = cfg->alpha;
    fa.alpha_neg = (float)(1.0 - (double)cfg->alpha);
    fa.loss_weight = cfg->loss_weight_cls;
    ba.sums = nullptr; ba.gin = grad_result; ba.res = result;
    for (int k = 0; k < 4; ++k) { ba.means[k] = g->means[k]; ba.stds[k] = g->stds[k]; }
    ba.beta = cfg->beta; ba.lw_bbox = cfg->loss_weight_bbox; ba.lw_iou = 1.0f;
    ba.attach = cfg->attach_iou_target ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)fa.lv.blk_off[L], fgrid = (unsigned)fa.lv.fblk_off[L];
    if (dtype == IA_F32) {
        hipLaunchKernelGGL((k_focal_ml<float, true>), dim3(fgrid), dim3(64), 0, s, fa);
        hipLaunchKernelGGL((k_box_ml<float, true>), dim3(grid), dim3(256), 0, s, ba);
    } else if (dtype == IA_BF16) {
        hipLaunchKernelGGL((k_focal_ml<uint16_t, true>), dim3(fgrid), dim3(64), 0, s, fa);
        hipLaunchKernelGGL((k_box_ml<uint16_t, true>), dim3(grid), dim3(256), 0, s, ba);
    } else return IA_E_ARG;
    return hip_status(hipGetLastError());
}


namespace ia {
static int fill_levels_nhwc(const ia_head_geom *g, int B, NhwcLevels &lv)
{
    if (!g || B < 1) return IA_E_ARG;
    if (g->num_levels < 1 || g->num_levels > IA_MAX_LEVELS) return IA_E_ARG;
    if (g->num_anchors < 1 || g->num_anchors > IA_MAX_ANCHORS || g->num_classes < 4 ||
        (g->num_classes & 3))
        return IA_E_ARG;                                  // class quads: C % 4 == 0
    if (g->cls_activation != IA_CLS_SIGMOID) return IA_E_ARG;    // sigmoid focal loss only
    if ((int64_t)g->num_anchors * (g->num_classes / 4) > 8192) return IA_E_ARG;   // float-reciprocal division
    lv.L = g->num_levels; lv.B = B; lv.A = g->num_anchors; lv.C = g->num_classes;
    int64_t foff = 0, boff = 0;
    lv.fblk_off[0] = lv.bblk_off[0] = 0;
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        const bool on = l < lv.L;
        if (on && (g->H[l] < 1 || g->W[l] < 1)) return IA_E_ARG;
        lv.H[l] = on ? g->H[l] : 0; lv.W[l] = on ? g->W[l] : 0; lv.stride[l] = on ? g->stride[l] : 0;
    }
    for (int o = 0; o < IA_MAX_LEVELS; ++o) {
        if (o < lv.L) {
            const int l = lv.L - 1 - o;
            const int64_t anchors = (int64_t)B * lv.H[l] * lv.W[l] * lv.A;
            foff += (anchors * (lv.C / 4) + kFocalChunks - 1) / kFocalChunks;
            boff += (anchors + 255) / 256;
            if (foff > 2147483647LL || boff > 2147483647LL) return IA_E_ARG;
        }
        lv.fblk_off[o + 1] = (int32_t)foff;
        lv.bblk_off[o + 1] = (int32_t)boff;
    }
    return 0;
}

static int check_strides(const NhwcLevels &lv, const ia_level_pix_strides *st, const ia_level_ptrs *p)
{
    for (int l = 0; l < lv.L; ++l) {
        if (st->cls[l] < (int64_t)lv.A * lv.C || st->reg[l] < (int64_t)lv.A * 4 || st->iou[l] < lv.A)
            return IA_E_ARG;
        if ((st->cls[l] & 3) || (st->reg[l] & 3)) return IA_E_ARG;          // 16-byte pieces
        if (((uintptr_t)p->cls[l] & 15u) || ((uintptr_t)p->reg[l] & 15u) || ((uintptr_t)p->iou[l] & 3u))
            return IA_E_ARG;
    }
    return 0;
}
}  // namespace ia

int ia_head_loss_fwd_nhwc(const ia_head_geom *g, const ia_level_ptrs *p,
                          const ia_level_pix_strides *strides, int batch, const ia_head_targets *t,
                          const ia_head_loss_cfg *cfg, void *workspace, size_t workspace_bytes,
                          float *result, void *stream)
{
    using namespace ia;
    if (!p || !strides || !t || !cfg || !workspace || !result || ((uintptr_t)workspace & 255u))
        return IA_E_ARG;
    FocalNhwcArgs fa;
    int rc = fill_levels_nhwc(g, batch, fa.lv);
    if (rc) return rc;
    const int L = fa.lv.L;
    if (workspace_bytes < sizeof(double) * kNumLoss * (size_t)L * IA_LOSS_SLOTS) return IA_E_WORKSPACE;
    if (cfg->gamma != 2.0f || !(cfg->beta > 0.0f)) return IA_E_ARG;
    for (int l = 0; l < L; ++l)
        if (!p->cls[l] || !p->reg[l] || !p->iou[l] || !t->labels[l] || !t->label_weights[l] ||
            !t->bbox_targets[l] || !t->bbox_weights[l])
            return IA_E_ARG;
    if ((rc = check_strides(fa.lv, strides, p))) return rc;
    BoxNhwcArgs ba;
    ba.lv = fa.lv;
    memcpy(ba.ba.v, g->base_anchors, sizeof(ba.ba.v));
    double *sums = static_cast<double *>(workspace);
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        const bool on = l < L;
        fa.cls[l] = on ? (const float *)p->cls[l] : nullptr;
        fa.ps_cls[l] = on ? strides->cls[l] : 0; fa.ps_grad[l] = 0; fa.grad[l] = nullptr;
        fa.labels[l] = on ? t->labels[l] : nullptr; fa.lw[l] = on ? t->label_weights[l] : nullptr;
        ba.reg[l] = on ? (const float *)p->reg[l] : nullptr; ba.iou[l] = on ? (const float *)p->iou[l] : nullptr;
        ba.ps_reg[l] = on ? strides->reg[l] : 0; ba.ps_iou[l] = on ? strides->iou[l] : 0;
        ba.pg_reg[l] = ba.pg_iou[l] = 0;
        ba.bt[l] = on ? t->bbox_targets[l] : nullptr; ba.bw[l] = on ? t->bbox_weights[l] : nullptr;
        ba.g_reg[l] = ba.g_iou[l] = nullptr;
    }
    fa.sums = sums; fa.gin = fa.res = nullptr;
    fa.big_logits = cfg->exact_large_logits ? 1 : 0;
    fa.alpha_pos = cfg->alpha;
    fa.alpha_neg = (float)(1.0 - (double)cfg->alpha);
    fa.loss_weight = cfg->loss_weight_cls;
    ba.sums = sums; ba.gin = ba.res = nullptr;
    for (int k = 0; k < 4; ++k) { ba.means[k] = g->means[k]; ba.stds[k] = g->stds[k]; }
    ba.beta = cfg->beta; ba.lw_bbox = cfg->loss_weight_bbox; ba.lw_iou = 1.0f;
    ba.attach = cfg->attach_iou_target ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * kNumLoss * (size_t)L * IA_LOSS_SLOTS, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_focal_nhwc<false>), dim3((unsigned)fa.lv.fblk_off[L]), dim3(256), 0, s, fa);
    for (int l = 0; l < IA_MAX_LEVELS; ++l) ba.g_pad[l] = 0;
    hipLaunchKernelGGL((k_box_nhwc<false>), dim3((unsigned)fa.lv.bblk_off[L]), dim3(256), 0, s, ba);
    FinArgs f;
    f.sums = sums; f.counts = t->counts; f.avg_dev = t->avg_factor_dev; f.avg_host = t->avg_factor;
    if (!f.counts && !f.avg_dev && !(f.avg_host > 0.0f)) return IA_E_ARG;
    f.lw[0] = cfg->loss_weight_cls; f.lw[1] = cfg->loss_weight_bbox; f.lw[2] = 1.0f;
    f.L = L; f.B = batch; f.res = result;
    hipLaunchKernelGGL(k_headloss_finalize, dim3(1), dim3(64), 0, s, f);
    return hip_status(hipGetLastError());
}

int ia_head_loss_bwd_nhwc(const ia_head_geom *g, const ia_level_ptrs *p,
                          const ia_level_pix_strides *strides, int batch, const ia_head_targets *t,
                          const ia_head_loss_cfg *cfg, const float *result, const float *grad_result,
                          const ia_level_ptrs *grads, const ia_level_pix_strides *grad_strides,
                          void *stream)
{
    using namespace ia;
    if (!p || !strides || !t || !cfg || !result || !grad_result || !grads || !grad_strides)
        return IA_E_ARG;
    FocalNhwcArgs fa;
    int rc = fill_levels_nhwc(g, batch, fa.lv);
    if (rc) return rc;
    const int L = fa.lv.L;
    if (cfg->gamma != 2.0f || !(cfg->beta > 0.0f)) return IA_E_ARG;
    for (int l = 0; l < L; ++l)
        if (!p->cls[l] || !p->reg[l] || !p->iou[l] || !t->labels[l] || !t->label_weights[l] ||
            !t->bbox_targets[l] || !t->bbox_weights[l] || !grads->cls[l] || !grads->reg[l] ||
            !grads->iou[l])
            return IA_E_ARG;
    if ((rc = check_strides(fa.lv, strides, p)) || (rc = check_strides(fa.lv, grad_strides, grads)))
        return rc;
    BoxNhwcArgs ba;
    ba.lv = fa.lv;
    memcpy(ba.ba.v, g->base_anchors, sizeof(ba.ba.v));
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        const bool on = l < L;
        fa.cls[l] = on ? (const float *)p->cls[l] : nullptr;
        fa.ps_cls[l] = on ? strides->cls[l] : 0;
        fa.grad[l] = on ? (float *)grads->cls[l] : nullptr;
        fa.ps_grad[l] = on ? grad_strides->cls[l] : 0;
        fa.labels[l] = on ? t->labels[l] : nullptr; fa.lw[l] = on ? t->label_weights[l] : nullptr;
        ba.reg[l] = on ? (const float *)p->reg[l] : nullptr; ba.iou[l] = on ? (const float *)p->iou[l] : nullptr;
        ba.ps_reg[l] = on ? strides->reg[l] : 0; ba.ps_iou[l] = on ? strides->iou[l] : 0;
        ba.pg_reg[l] = on ? grad_strides->reg[l] : 0; ba.pg_iou[l] = on ? grad_strides->iou[l] : 0;
        ba.bt[l] = on ? t->bbox_targets[l] : nullptr; ba.bw[l] = on ? t->bbox_weights[l] : nullptr;
        ba.g_reg[l] = on ? (float *)grads->reg[l] : nullptr;
        ba.g_iou[l] = on ? (float *)grads->iou[l] : nullptr;
        // both gradients in one pixel row that STARTS at reg (the caller says so: cfg->grad_rows_start_at_reg;
        // a row [X | reg | iou | pad] looks the same from here, and zero-filling behind iou would run into the
        // next pixel's X): the channels left up to the row's end get their zero gradient here
        ba.g_pad[l] = 0;
        if (on && cfg->grad_rows_start_at_reg) {
            if (ba.g_iou[l] != ba.g_reg[l] + 4 * fa.lv.A || ba.pg_reg[l] != ba.pg_iou[l] ||
                ba.pg_reg[l] < 5 * fa.lv.A || ba.pg_reg[l] - 5 * fa.lv.A > 64)
                return IA_E_ARG;
            ba.g_pad[l] = (int32_t)(ba.pg_reg[l] - 5 * fa.lv.A);
        }
    }
    fa.sums = nullptr; fa.gin = grad_result; fa.res = result;
    fa.big_logits = 0;
    fa.alpha_pos = cfg->alpha;
    fa.alpha_neg = (float)(1.0 - (double)cfg->alpha);
    fa.loss_weight = cfg->loss_weight_cls;
    ba.sums = nullptr; ba.gin = grad_result; ba.res = result;
    for (int k = 0; k < 4; ++k) { ba.means[k] = g->means[k]; ba.stds[k] = g->stds[k]; }
    ba.beta = cfg->beta; ba.lw_bbox = cfg->loss_weight_bbox; ba.lw_iou = 1.0f;
    ba.attach = cfg->attach_iou_target ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((k_focal_nhwc<true>), dim3((unsigned)fa.lv.fblk_off[L]), dim3(256), 0, s, fa);
    hipLaunchKernelGGL((k_box_nhwc<true>), dim3((unsigned)fa.lv.bblk_off[L]), dim3(256), 0, s, ba);
    return hip_status(hipGetLastError());
}

}  // extern "C"
