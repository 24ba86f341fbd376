// Training-loss kernels of IoUawareRetinaHead.loss_single
// (reference iou_aware_retina_head.py:221-313), computed on the NCHW head
// outputs without the permute/reshape copies, the (N,80) int64 one-hot
// (anchor_target.py:247-254) or the (N,80) expanded weight the reference
// materialises:
//   focal      FocalLoss / py_sigmoid_focal_loss (losses.py:226-247,279-303)
//   smooth-L1  weighted_smoothl1                 (losses.py:385-411)
//   IoU-BCE    delta2bbox x2 + aligned IoU + BCE (head :256-259,:276-281;
//                                                 geometry.py:34-47; losses.py:460-480)
//   focal-op   the reference CUDA op's formula   (sigmoid_focal_loss_cuda.cu:23-105)
//
// All are streaming kernels; the dominant one (focal) re-uses the row-max
// kernel's tiling: a wavefront owns one anchor and 256 consecutive positions,
// so every class-plane access is a contiguous 1 KiB segment, and the label /
// weight of an anchor are read once for all C classes.  Sums are reduced in
// fp64 (wave shuffle -> LDS -> one fp64 atomic per workgroup).
#include "ia_loss.hpp"

namespace ia {

// ------------------------------------------------------------------ focal
struct FocalArgs {
    const void *cls;
    const int64_t *labels;
    const float *label_weights;
    double *loss_sum;
    float *grad;
    int32_t B, A, C, HW;
    int32_t cchunk;          // classes per wavefront (the class range is split over blockIdx.y)
    float gamma, alpha_pos, alpha_neg, gscale;
    const float *gscale_dev;
    // IoU-balanced variant (losses.py:309-374): per-anchor IoU, exponent, and (backward) the
    // normalizer S1 / (S2 + 1e-6) as a device scalar
    const float *anchor_iou;
    const float *pos_scale_dev;
    float eta;
};

// one element of py_sigmoid_focal_loss (losses.py:232-236) and its derivative w.r.t. the logit.
// One exp, one reciprocal, one log: e = exp(-|x|); sigmoid and 1-sigmoid are e/(1+e), 1/(1+e).
template <bool BWD, bool GAMMA2>
__device__ __forceinline__ float focal_elem(float x, bool t, float at, float gamma, float gs)
{
    const float ax = __builtin_fabsf(x);
    const float e = fastm::exp_(-ax);
    const float inv = fastm::rcp_(1.0f + e);
    const float hi = inv, lo = e * inv;                  // sigmoid(|x|), sigmoid(-|x|)
    const bool pos = x >= 0.0f;
    const float pr = pos ? hi : lo;                      // p
    const float qr = pos ? lo : hi;                      // 1 - p  (no cancellation)
    const float sp = fastm::log_(1.0f + e);              // log(1 + exp(-|x|))
    const float bce = (t ? (pos ? 0.0f : ax) : (pos ? ax : 0.0f)) + sp;
    const float pt = t ? qr : pr;
    float mod, dmod;
    if (GAMMA2) { mod = pt * pt; dmod = 2.0f * pt; }      // compile-time: no speculated log/exp
    else {
        const float lg = __builtin_amdgcn_logf(pt);      // log2
        mod = __builtin_amdgcn_exp2f(gamma * lg);
        dmod = gamma * __builtin_amdgcn_exp2f((gamma - 1.0f) * lg);
    }
    if (!BWD) return bce * (at * mod);
    const float dbce = t ? -qr : pr;                     // p - t
    const float dpt = t ? -(pr * qr) : (pr * qr);
    return (dbce * (at * mod) + (bce * at) * (dmod * dpt)) * gs;
}

template <typename T> struct LPack;
template <> struct LPack<float> {
    static constexpr int N = 4;
    using V = float4;
    static __device__ __forceinline__ void unpack(const V &q, float (&v)[4]) { v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
};
template <> struct LPack<uint16_t> {
    static constexpr int N = 4;
    using V = ushort4;
    static __device__ __forceinline__ void unpack(const V &q, float (&v)[4])
    {
        v[0] = bf16_to_f32(q.x); v[1] = bf16_to_f32(q.y); v[2] = bf16_to_f32(q.z); v[3] = bf16_to_f32(q.w);
    }
};

// One wavefront per (image, anchor, tile of 256 positions) -- the row-max kernel's tiling:
// every class-plane access is a contiguous 1 KiB segment, the label / weight of an anchor are
// read once for all C classes.
// BAL: the positive element of a positive anchor is weighted by iou^eta * normalizer; the forward
// accumulates S0 (t = 0 elements), S1 = sum_pos loss, S2 = sum_pos loss * iou^eta in
// loss_sum[0:64], [64:128], [128:192].
template <typename T, bool BWD, bool GAMMA2, bool BAL>
__global__ void __launch_bounds__(64) k_focal(FocalArgs a)
{
    const int lane = threadIdx.x;
    const int A = a.A, C = a.C, HW = a.HW;
    const int tiles = (HW + 255) / 256;
    int bid = blockIdx.x;
    const int tile = bid % tiles; bid /= tiles;
    const int an = bid % A;
    const int b = bid / A;
    const int p0 = tile * 256;
    const int cbeg = blockIdx.y * a.cchunk;
    const int cend = (cbeg + a.cchunk < C) ? (cbeg + a.cchunk) : C;
    const T *cls = static_cast<const T *>(a.cls) + ((size_t)b * A + an) * C * HW;
    float *grad = BWD ? a.grad + ((size_t)b * A + an) * C * HW : nullptr;
    const float gs = BWD ? eff_scale(a.gscale, a.gscale_dev) : 1.0f;
    const bool vec = (HW & 3) == 0;
    int pos[4], pc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pos[j] = vec ? (p0 + lane * 4 + j) : (p0 + lane + 64 * j);
        pc[j] = (pos[j] < HW) ? pos[j] : (vec ? (HW - 4 + j) : (HW - 1));   // clamped: loads unconditional
    }
    int lab[4];
    float at_pos[4], at_neg[4], iw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t n = ((size_t)b * HW + pc[j]) * A + an;
        lab[j] = (int)a.labels[n];
        const float w0 = (pos[j] < HW) ? a.label_weights[n] : 0.0f;          // padding lanes weigh 0
        at_pos[j] = a.alpha_pos * w0;
        at_neg[j] = a.alpha_neg * w0;
        iw[j] = 1.0f;
        if (BAL) {                                         // (t * iou)^eta, times the normalizer
            iw[j] = __builtin_amdgcn_exp2f(a.eta * __builtin_amdgcn_logf(a.anchor_iou[n]));
            if (BWD) iw[j] *= a.pos_scale_dev[0];
        }
    }
    float acc = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
    auto account = [&](float &o, bool t, int j) {
        if (!BAL) { acc += o; return; }
        if (BWD) { o = t ? o * iw[j] : o; return; }
        acc += t ? 0.0f : o;
        acc1 += t ? o : 0.0f;
        acc2 += t ? o * iw[j] : 0.0f;
    };
    if (vec) {
        const T *src = cls + pc[0];
        constexpr int K = 8;                               // class planes in flight per wavefront
        using V = typename LPack<T>::V;
        for (int c0 = cbeg; c0 < cend; c0 += K) {
            V q[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {                  // all loads first: independent, unconditional
                const int c = (c0 + i < cend) ? (c0 + i) : (cend - 1);
                q[i] = *reinterpret_cast<const V *>(src + (size_t)c * HW);
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int c = c0 + i;
                if (c < cend) {                            // wave-uniform
                    float v[4], o[4];
                    LPack<T>::unpack(q[i], v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool t = lab[j] == c + 1;
                        o[j] = focal_elem<BWD, GAMMA2>(v[j], t, t ? at_pos[j] : at_neg[j], a.gamma, gs);
                        account(o[j], t, j);
                    }
                    if (BWD && pos[0] < HW)
                        *reinterpret_cast<float4 *>(grad + (size_t)c * HW + pc[0]) =
                            make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    } else {
#pragma unroll 2
        for (int c = cbeg; c < cend; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = load_f32<T>(cls + (size_t)c * HW + pc[j]);
                const bool t = lab[j] == c + 1;
                float o = focal_elem<BWD, GAMMA2>(x, t, t ? at_pos[j] : at_neg[j], a.gamma, gs);
                account(o, t, j);
                if (BWD && pos[j] < HW) grad[(size_t)c * HW + pos[j]] = o;
            }
        }
    }
    if (!BWD) {
        double d = wave_sum((double)acc);
        const int slot = (blockIdx.x + blockIdx.y) & (IA_LOSS_SLOTS - 1);
        if (lane == 0) atomicAdd(a.loss_sum + slot, d);
        if (BAL && __ballot(acc1 != 0.0f || acc2 != 0.0f)) {         // positives are rare
            const double d1 = wave_sum((double)acc1), d2 = wave_sum((double)acc2);
            if (lane == 0) {
                atomicAdd(a.loss_sum + IA_LOSS_SLOTS + slot, d1);
                atomicAdd(a.loss_sum + 2 * IA_LOSS_SLOTS + slot, d2);
            }
        }
    }
}

static int launch_focal(bool bwd, const void *cls, int dtype, const int64_t *labels,
                        const float *lw, int B, int A, int C, int HW, float gamma, float alpha,
                        float gscale, const float *gscale_dev, double *loss_sum, float *grad,
                        hipStream_t s, const float *anchor_iou = nullptr, float eta = 0.0f,
                        const float *pos_scale_dev = nullptr)
{
    const bool bal = anchor_iou != nullptr;
    if (bal && (!(eta > 0.0f) || (bwd && !pos_scale_dev))) return IA_E_ARG;
    if (!cls || !labels || !lw || B < 1 || A < 1 || A > IA_MAX_ANCHORS || C < 1 || HW < 1)
        return IA_E_ARG;
    if (bwd ? !grad : !loss_sum) return IA_E_ARG;
    FocalArgs a;
    a.cls = cls; a.labels = labels; a.label_weights = lw; a.loss_sum = loss_sum; a.grad = grad;
    a.B = B; a.A = A; a.C = C; a.HW = HW; a.gamma = gamma; a.gscale = gscale; a.gscale_dev = gscale_dev;
    a.alpha_pos = alpha;
    a.alpha_neg = (float)(1.0 - (double)alpha);   // python: (1 - alpha) in double, then fp32
    a.anchor_iou = anchor_iou; a.eta = eta; a.pos_scale_dev = pos_scale_dev;
    const int64_t blocks = (int64_t)B * A * ((HW + 255) / 256);
    if (blocks > 2147483647LL) return IA_E_ARG;
    // Small levels: split the class range over blockIdx.y so that there are >= ~2 wavefronts per
    // SIMD (a lone wavefront runs the dependent exp/rcp/log chain at issue latency).  Large levels
    // are NOT split: measured on MI355X (B=4, P3) 1/2/4/5/10 splits -> 49/50/63/64/89 us.
    int split = (int)((2048 + blocks - 1) / blocks);
    if (split < 1) split = 1;
    if (split > (C + 7) / 8) split = (C + 7) / 8;
    a.cchunk = ((C + split - 1) / split + 7) / 8 * 8;
    const int ny = (C + a.cchunk - 1) / a.cchunk;
    dim3 block(64), grid((unsigned)blocks, (unsigned)ny);
    const bool g2 = gamma == 2.0f;
#define IA_FOCAL(T, BAL)                                                                   \
    do {                                                                                       \
        if (bwd && g2) hipLaunchKernelGGL((k_focal<T, true, true, BAL>), grid, block, 0, s, a);    \
        else if (bwd) hipLaunchKernelGGL((k_focal<T, true, false, BAL>), grid, block, 0, s, a);    \
        else if (g2) hipLaunchKernelGGL((k_focal<T, false, true, BAL>), grid, block, 0, s, a);     \
        else hipLaunchKernelGGL((k_focal<T, false, false, BAL>), grid, block, 0, s, a);            \
    } while (0)
    if (dtype == IA_F32) { if (bal) IA_FOCAL(float, true); else IA_FOCAL(float, false); }
    else if (dtype == IA_BF16) { if (bal) IA_FOCAL(uint16_t, true); else IA_FOCAL(uint16_t, false); }
    else return IA_E_ARG;
#undef IA_FOCAL
    return hip_status(hipGetLastError());
}

// ------------------------------------------------------------------ smooth L1
struct SmoothArgs {
    const void *pred;
    const float *target;
    const float *weight;
    double *loss_sum;
    float *grad;
    int32_t B, A, HW;
    float beta, gscale;
    const float *gscale_dev;
    const float *anchor_iou;      // IoU-balanced variant (losses.py:416-458): weight * iou^delta
    float delta;
};

// thread = (image, anchor, position); the 4 delta planes of an anchor are read coalesced along
// the position, target / weight rows (16 B) from the (B, N_l, 4) arrays.  32-bit index math.
template <typename T, bool BWD>
__global__ void __launch_bounds__(256) k_smooth_l1(SmoothArgs a)
{
    __shared__ double red[16];
    const int A = a.A, HW = a.HW;
    const int tiles = (HW + 255) / 256;
    int bid = blockIdx.x;
    const int tile = bid % tiles; bid /= tiles;
    const int an = bid % A;
    const int b = bid / A;
    const int p = tile * 256 + threadIdx.x;
    float acc = 0.0f;
    if (p < HW) {
        const size_t n = ((size_t)b * HW + p) * A + an;
        const float4 tg = reinterpret_cast<const float4 *>(a.target)[n];
        const float4 wt = reinterpret_cast<const float4 *>(a.weight)[n];
        const float tv[4] = {tg.x, tg.y, tg.z, tg.w};
        float wv[4] = {wt.x, wt.y, wt.z, wt.w};
        if (a.anchor_iou) {                                 // exact math: equals the oracle's bits
            const float pw = powf_pos_(a.anchor_iou[n], a.delta);
#pragma unroll
            for (int k = 0; k < 4; ++k) wv[k] = wv[k] * pw;
        }
        const size_t e0 = (((size_t)b * A + an) * 4) * HW + p;
        const float gs = BWD ? eff_scale(a.gscale, a.gscale_dev) : 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t e = e0 + (size_t)k * HW;
            float df = load_f32<T>(static_cast<const T *>(a.pred) + e) - tv[k];
            float d = __builtin_fabsf(df);
            if (BWD) {
                float sgn = (df > 0.0f) ? 1.0f : ((df < 0.0f) ? -1.0f : 0.0f);
                float g = (d < a.beta) ? df / a.beta : sgn;
                a.grad[e] = (g * wv[k]) * gs;
            } else {
                float l = (d < a.beta) ? ((0.5f * d) * d) / a.beta : d - 0.5f * a.beta;
                acc += l * wv[k];
            }
        }
    }
    if (!BWD) block_sum_to((double)acc, a.loss_sum, red);
}

static int launch_smooth(bool bwd, const void *pred, int dtype, const float *target,
                         const float *weight, int B, int A, int HW, float beta, float gscale,
                         const float *gscale_dev, double *loss_sum, float *grad, hipStream_t s,
                         const float *anchor_iou = nullptr, float delta = 0.0f)
{
    if (!pred || !target || !weight || B < 1 || A < 1 || HW < 1 || !(beta > 0.0f)) return IA_E_ARG;
    if (bwd ? !grad : !loss_sum) return IA_E_ARG;
    SmoothArgs a;
    a.pred = pred; a.target = target; a.weight = weight; a.loss_sum = loss_sum; a.grad = grad;
    a.B = B; a.A = A; a.HW = HW; a.beta = beta; a.gscale = gscale; a.gscale_dev = gscale_dev;
    a.anchor_iou = anchor_iou; a.delta = delta;
    const int64_t blocks = (int64_t)B * A * ((HW + 255) / 256);
    if (blocks > 2147483647LL) return IA_E_ARG;
    unsigned grid = (unsigned)blocks;
    if (dtype == IA_F32) {
        if (bwd) hipLaunchKernelGGL((k_smooth_l1<float, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_smooth_l1<float, false>), dim3(grid), dim3(256), 0, s, a);
    } else if (dtype == IA_BF16) {
        if (bwd) hipLaunchKernelGGL((k_smooth_l1<uint16_t, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_smooth_l1<uint16_t, false>), dim3(grid), dim3(256), 0, s, a);
    } else return IA_E_ARG;
    return hip_status(hipGetLastError());
}

// ------------------------------------------------------------------ IoU target + BCE
struct IouBceArgs {
    const void *bbox_pred;
    const void *iou_pred;
    const float *bbox_targets;
    const float *bbox_weights;
    float *iou_target;
    double *loss_sum;
    float *grad_iou_pred;
    float *grad_bbox_pred;
    float base[IA_MAX_ANCHORS][4];
    float means[4], stds[4];
    int32_t B, A, H, W, stride;
    float gscale;
    const float *gscale_dev;
};

template <typename T, bool BWD>
__global__ void __launch_bounds__(256) k_iou_bce(IouBceArgs a)
{
    __shared__ double red[16];
    const int A = a.A, W = a.W, HW = a.H * a.W;
    double acc = 0.0;
    const float gs = BWD ? eff_scale(a.gscale, a.gscale_dev) : 1.0f;
    const int tiles = (HW + 255) / 256;
    int bid = blockIdx.x;
    const int tile = bid % tiles; bid /= tiles;
    const int an = bid % A;
    const size_t b = (size_t)(bid / A);
    const int p = tile * 256 + threadIdx.x;
    if (p < HW) {
        const size_t ba = b * A + an;
        const size_t e = ba * HW + p;
        const size_t n = (b * HW + p) * A + an;
        const int y = p / W, x = p - y * W;
        const float sx = (float)(x * a.stride), sy = (float)(y * a.stride);
        const float anc[4] = {a.base[an][0] + sx, a.base[an][1] + sy, a.base[an][2] + sx,
                              a.base[an][3] + sy};
        const T *bp = static_cast<const T *>(a.bbox_pred) + ba * 4 * HW + p;
        const float dp[4] = {load_f32<T>(bp), load_f32<T>(bp + (size_t)HW),
                             load_f32<T>(bp + (size_t)2 * HW), load_f32<T>(bp + (size_t)3 * HW)};
        const float4 tq = reinterpret_cast<const float4 *>(a.bbox_targets)[n];
        const float dt[4] = {tq.x, tq.y, tq.z, tq.w};
        const IouElem q = iou_target_elem(anc, dp, dt, a.means, a.stds);
        const float t = q.t;
        float xl = load_f32<T>(static_cast<const T *>(a.iou_pred) + e);
        float wt = a.bbox_weights[4 * n];
        if (!BWD) {
            if (a.iou_target) a.iou_target[n] = t;
            acc += (double)(bce_logits_(xl, t) * wt);
        } else {
            if (a.grad_iou_pred) a.grad_iou_pred[e] = ((sigmoidf_(xl) - t) * wt) * gs;
            if (a.grad_bbox_pred) {
                float gv[4];
                iou_bce_box_grad(q, xl, wt, gs, a.stds, gv);
                float *go = a.grad_bbox_pred + ba * 4 * HW + p;
                go[0] = gv[0];
                go[(size_t)HW] = gv[1];
                go[(size_t)2 * HW] = gv[2];
                go[(size_t)3 * HW] = gv[3];
            }
        }
    }
    if (!BWD) block_sum_to(acc, a.loss_sum, red);
}

static int launch_iou_bce(bool bwd, const ia_head_geom *g, int level, const void *bbox_pred,
                          const void *iou_pred, int dtype, const float *bt, const float *bw, int B,
                          float gscale, const float *gscale_dev, float *iou_target,
                          double *loss_sum, float *g_iou, float *g_box, hipStream_t s)
{
    if (!g || level < 0 || level >= g->num_levels || !bbox_pred || !iou_pred || !bt || !bw || B < 1)
        return IA_E_ARG;
    if (g->num_anchors < 1 || g->num_anchors > IA_MAX_ANCHORS) return IA_E_ARG;
    if (g->layout != IA_LAYOUT_NCHW) return IA_E_ARG;            // training kernels: NCHW only
    if (!bwd && !loss_sum) return IA_E_ARG;
    IouBceArgs a;
    a.bbox_pred = bbox_pred; a.iou_pred = iou_pred; a.bbox_targets = bt; a.bbox_weights = bw;
    a.iou_target = iou_target; a.loss_sum = loss_sum; a.grad_iou_pred = g_iou;
    a.grad_bbox_pred = g_box;
    for (int i = 0; i < IA_MAX_ANCHORS; ++i)
        for (int k = 0; k < 4; ++k) a.base[i][k] = g->base_anchors[level][i][k];
    for (int k = 0; k < 4; ++k) { a.means[k] = g->means[k]; a.stds[k] = g->stds[k]; }
    a.B = B; a.A = g->num_anchors; a.H = g->H[level]; a.W = g->W[level];
    a.stride = g->stride[level]; a.gscale = gscale; a.gscale_dev = gscale_dev;
    const int64_t blocks = (int64_t)B * a.A * ((a.H * a.W + 255) / 256);
    if (blocks > 2147483647LL) return IA_E_ARG;
    unsigned grid = (unsigned)blocks;
    if (dtype == IA_F32) {
        if (bwd) hipLaunchKernelGGL((k_iou_bce<float, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_iou_bce<float, false>), dim3(grid), dim3(256), 0, s, a);
    } else if (dtype == IA_BF16) {
        if (bwd) hipLaunchKernelGGL((k_iou_bce<uint16_t, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_iou_bce<uint16_t, false>), dim3(grid), dim3(256), 0, s, a);
    } else return IA_E_ARG;
    return hip_status(hipGetLastError());
}

// ------------------------------------------------------------------ focal op (CUDA-op formula)
struct FocalOpArgs {
    const void *logits;
    const int64_t *targets;
    const void *d_losses;
    void *out;
    int64_t total;
    int32_t C, dtype;        // IA_F32 / IA_BF16 / IA_F16 / IA_F64: storage type of logits, d_losses, out
    float gamma, alpha;
};

// the reference op is instantiated for float, double and half (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// sigmoid_focal_loss_cuda.cu:128,166) and evaluates expf / powf / logf -- single precision --
// whatever the storage type: load -> fp32 math -> one rounding at the store
__device__ __forceinline__ float focal_op_load(const void *p, int64_t i, int dtype)
{
    switch (dtype) {
    case IA_BF16: return bf16_to_f32(static_cast<const uint16_t *>(p)[i]);
    case IA_F16: return (float)static_cast<const _Float16 *>(p)[i];
    case IA_F64: return (float)static_cast<const double *>(p)[i];
    default: return static_cast<const float *>(p)[i];
    }
}

__device__ __forceinline__ void focal_op_store(void *p, int64_t i, int dtype, float v)
{
    switch (dtype) {
    case IA_BF16: {
        uint32_t u = to_bits(v);
        u = ((u & 0x7fffffffu) > 0x7f800000u) ? ((u >> 16) | 0x40u) : ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        static_cast<uint16_t *>(p)[i] = (uint16_t)u;
        break;
    }
    case IA_F16: static_cast<_Float16 *>(p)[i] = (_Float16)v; break;
    case IA_F64: static_cast<double *>(p)[i] = (double)v; break;
    default: static_cast<float *>(p)[i] = v;
    }
}

template <bool BWD>
__global__ void __launch_bounds__(256) k_focal_op(FocalOpArgs a)
{
    const float FLT_MIN_ = 1.17549435e-38f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / a.C;
        const int d = (int)(i - n * a.C);
        const int t = (int)a.targets[n];
        const float c1 = (t == d + 1) ? 1.0f : 0.0f;
        const float c2 = ((t >= 0) & (t != d + 1)) ? 1.0f : 0.0f;
        const float zn = 1.0f - a.alpha, zp = a.alpha;
        const float x = focal_op_load(a.logits, i, a.dtype);
        const float p = 1.0f / (1.0f + expf_(-x));
        const float pm = (p > FLT_MIN_) ? p : FLT_MIN_;
        const float xs = (x >= 0.0f) ? x : 0.0f;
        const float lg2 = (-1.0f * xs) - logf_(1.0f + expf_(x - 2.0f * xs));
        float r = 0.0f;
        if (!BWD) {
            float term1 = powf_pos_(1.0f - p, a.gamma) * logf_(pm);
            float term2 = powf_pos_(p, a.gamma) * lg2;
            r += -c1 * term1 * zp;
            r += -c2 * term2 * zn;
        } else {
            float term1 = powf_pos_(1.0f - p, a.gamma) * ((1.0f - p) - (p * a.gamma) * logf_(pm));
            float term2 = powf_pos_(p, a.gamma) * ((lg2 * (1.0f - p)) * a.gamma - p);
            r += -c1 * term1 * zp;
            r += -c2 * term2 * zn;
            r = r * focal_op_load(a.d_losses, i, a.dtype);
        }
        focal_op_store(a.out, i, a.dtype, r);
    }
}

static int launch_focal_op(bool bwd, const void *logits, const int64_t *targets,
                           const void *d_losses, int N, int C, float gamma, float alpha,
                           void *out, hipStream_t s, int dtype = IA_F32)
{
    if (N < 0 || C < 1) return IA_E_ARG;
    if (dtype != IA_F32 && dtype != IA_BF16 && dtype != IA_F16 && dtype != IA_F64) return IA_E_ARG;
    if (N == 0) return 0;
    if (!logits || !targets || !out || (bwd && !d_losses)) return IA_E_ARG;
    FocalOpArgs a;
    a.logits = logits; a.targets = targets; a.d_losses = d_losses; a.out = out;
    a.total = (int64_t)N * C; a.C = C; a.dtype = dtype; a.gamma = gamma; a.alpha = alpha;
    int64_t blocks = (a.total + 255) / 256;
    unsigned grid = (unsigned)(blocks > 4096 ? 4096 : blocks);
    if (bwd) hipLaunchKernelGGL(k_focal_op<true>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_focal_op<false>, dim3(grid), dim3(256), 0, s, a);
    return hip_status(hipGetLastError());
}

}  // namespace ia

// ------------------------------------------------------------------ C ABI (loss half)
extern "C" {

int ia_focal_loss_fwd(const void *cls, int dtype, const int64_t *labels, const float *lw, int B,
                      int A, int C, int HW, float gamma, float alpha, double *loss_sum, void *stream)
{
    return ia::launch_focal(false, cls, dtype, labels, lw, B, A, C, HW, gamma, alpha, 1.0f,
                            nullptr, loss_sum, nullptr, (hipStream_t)stream);
}
int ia_focal_loss_bwd(const void *cls, int dtype, const int64_t *labels, const float *lw, int B,
                      int A, int C, int HW, float gamma, float alpha, float gscale,
                      const float *gscale_dev, float *grad, void *stream)
{
    return ia::launch_focal(true, cls, dtype, labels, lw, B, A, C, HW, gamma, alpha, gscale,
                            gscale_dev, nullptr, grad, (hipStream_t)stream);
}
int ia_smooth_l1_fwd(const void *pred, int dtype, const float *target, const float *weight, int B,
                     int A, int HW, float beta, double *loss_sum, void *stream)
{
    return ia::launch_smooth(false, pred, dtype, target, weight, B, A, HW, beta, 1.0f, nullptr,
                             loss_sum, nullptr, (hipStream_t)stream);
}
int ia_smooth_l1_bwd(const void *pred, int dtype, const float *target, const float *weight, int B,
                     int A, int HW, float beta, float gscale, const float *gscale_dev, float *grad,
                     void *stream)
{
    return ia::launch_smooth(true, pred, dtype, target, weight, B, A, HW, beta, gscale, gscale_dev,
                             nullptr, grad, (hipStream_t)stream);
}
int ia_focal_loss_balanced_fwd(const void *cls, int dtype, const int64_t *labels, const float *lw,
                               const float *anchor_iou, int B, int A, int C, int HW, float gamma,
                               float alpha, float eta, double *loss_sums3, void *stream)
{
    if (!anchor_iou) return IA_E_ARG;
    return ia::launch_focal(false, cls, dtype, labels, lw, B, A, C, HW, gamma, alpha, 1.0f,
                            nullptr, loss_sums3, nullptr, (hipStream_t)stream, anchor_iou, eta,
                            nullptr);
}
int ia_focal_loss_balanced_bwd(const void *cls, int dtype, const int64_t *labels, const float *lw,
                               const float *anchor_iou, int B, int A, int C, int HW, float gamma,
                               float alpha, float eta, const float *normalizer_dev, float gscale,
                               const float *gscale_dev, float *grad, void *stream)
{
    if (!anchor_iou) return IA_E_ARG;
    return ia::launch_focal(true, cls, dtype, labels, lw, B, A, C, HW, gamma, alpha, gscale,
                            gscale_dev, nullptr, grad, (hipStream_t)stream, anchor_iou, eta,
                            normalizer_dev);
}
int ia_smooth_l1_balanced_fwd(const void *pred, int dtype, const float *target,
                              const float *weight, const float *anchor_iou, int B, int A, int HW,
                              float beta, float delta, double *loss_sum, void *stream)
{
    if (!anchor_iou) return IA_E_ARG;
    return ia::launch_smooth(false, pred, dtype, target, weight, B, A, HW, beta, 1.0f, nullptr,
                             loss_sum, nullptr, (hipStream_t)stream, anchor_iou, delta);
}
int ia_smooth_l1_balanced_bwd(const void *pred, int dtype, const float *target,
                              const float *weight, const float *anchor_iou, int B, int A, int HW,
                              float beta, float delta, float gscale, const float *gscale_dev,
                              float *grad, void *stream)
{
    if (!anchor_iou) return IA_E_ARG;
    return ia::launch_smooth(true, pred, dtype, target, weight, B, A, HW, beta, gscale, gscale_dev,
                             nullptr, grad, (hipStream_t)stream, anchor_iou, delta);
}
int ia_iou_bce_fwd(const ia_head_geom *g, int level, const void *bbox_pred, const void *iou_pred,
                   int dtype, const float *bt, const float *bw, int B, float *iou_target,
                   double *loss_sum, void *stream)
{
    return ia::launch_iou_bce(false, g, level, bbox_pred, iou_pred, dtype, bt, bw, B, 1.0f, nullptr,
                              iou_target, loss_sum, nullptr, nullptr, (hipStream_t)stream);
}
int ia_iou_bce_bwd(const ia_head_geom *g, int level, const void *bbox_pred, const void *iou_pred,
                   int dtype, const float *bt, const float *bw, int B, float gscale,
                   const float *gscale_dev, float *g_iou, float *g_box, void *stream)
{
    return ia::launch_iou_bce(true, g, level, bbox_pred, iou_pred, dtype, bt, bw, B, gscale,
                              gscale_dev, nullptr, nullptr, g_iou, g_box, (hipStream_t)stream);
}
int ia_sigmoid_focal_loss_fwd(const float *logits, const int64_t *targets, int N, int C,
                              float gamma, float alpha, float *losses, void *stream)
{
    return ia::launch_focal_op(false, logits, targets, nullptr, N, C, gamma, alpha, losses,
                               (hipStream_t)stream);
}
int ia_sigmoid_focal_loss_bwd(const float *logits, const int64_t *targets, const float *d_losses,
                              int N, int C, float gamma, float alpha, float *d_logits, void *stream)
{
    return ia::launch_focal_op(true, logits, targets, d_losses, N, C, gamma, alpha, d_logits,
                               (hipStream_t)stream);
}

int ia_sigmoid_focal_loss_fwd_dt(const void *logits, int dtype, const int64_t *targets, int N, int C,
                                 float gamma, float alpha, void *losses, void *stream)
{
    return ia::launch_focal_op(false, logits, targets, nullptr, N, C, gamma, alpha, losses,
                               (hipStream_t)stream, dtype);
}
int ia_sigmoid_focal_loss_bwd_dt(const void *logits, int dtype, const int64_t *targets,
                                 const void *d_losses, int N, int C, float gamma, float alpha,
                                 void *d_logits, void *stream)
{
    return ia::launch_focal_op(true, logits, targets, d_losses, N, C, gamma, alpha, d_logits,
                               (hipStream_t)stream, dtype);
}

}  // extern "C"
