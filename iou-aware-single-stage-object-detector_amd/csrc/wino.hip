// Winograd F(4x4, 3x3) transforms for the 3x3 / stride-1 / pad-1 convolutions of the head
// (reference iou_aware_retina_head.py:171-219: 4+4 tower convs 256->256, retina_cls 256->720,
// retina_reg 256->36, retina_iou 256->9, weights shared by the five pyramid levels) and of the
// FPN outputs.  Those convolutions are 63 % of the network's multiply-adds at 800x1344; as direct
// / implicit-GEMM fp32 convolutions they run at ~110-125 TFLOP/s of the 157 TFLOP/s MFMA peak, so
// the only large lever left is the number of multiplications: F(4x4,3x3) needs 36 per 16 outputs
// instead of 144.
//
// Split of work:
//   k_wino_in   (this file)   d (6x6 input patch per tile, zero padded) -> V = B^T d B, scattered
//                             as 36 matrices V[k] of (tiles x Cin); ALL pyramid levels of the
//                             batch form one tile list, so the shared-weight head needs ONE
//                             batched GEMM per layer instead of five small ones;
//   batched GEMM              M[k] = V[k] (tiles x Cin) . U[k] (Cin x Cout), 36 (x groups) plain
//                             fp32 GEMMs -> rocBLAS / hipBLASLt through torch.bmm (library GEMM);
//   k_wino_out  (this file)   Y = A^T M A (4x4 outputs per tile) + bias (+ ReLU), written straight
//                             into the channels-last activation / head-output tensors.
// Activations are channels-last fp32: a wavefront handles one tile x 256 channels, 16 bytes per
// lane, so every load / store instruction moves one contiguous 1 KiB pixel row.  Both kernels
// are HBM-bound streams (36 x 16 B in, 36 x 16 B out per lane; 36 in, 16 out).
// Transform matrices: Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks"
// (interpolation points 0, +-1, +-2, inf); weights are transformed once on the host side
// (U = G g G^T in fp64, iouaware/winograd.py).
#include <string.h>
#include "ia_internal.hpp"
#include "ia_wino.hpp"

namespace ia {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
// streaming accesses: V is written once and read by the GEMM much later, M is read exactly once
__device__ __forceinline__ float4 load_nt(const float *p)
{
    const f32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p));
    return make_float4(q.x, q.y, q.z, q.w);
}
__device__ __forceinline__ void store_nt(float *p, const float4 &v)
{
    f32x4_t q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
    __builtin_nontemporal_store(q, reinterpret_cast<f32x4_t *>(p));
}
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 operator+(const float4 &a, const float4 &b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(const float4 &a, const float4 &b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float s, const float4 &a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }

// one line of B^T d (the same 6-point transform is applied to rows, then to columns)
__device__ __forceinline__ void bt6(const float4 (&d)[6], float4 (&o)[6])
{
    o[0] = (4.0f * d[0] - 5.0f * d[2]) + d[4];
    o[1] = (d[3] + d[4]) - 4.0f * (d[1] + d[2]);
    o[2] = 4.0f * (d[1] - d[2]) + (d[4] - d[3]);
    o[3] = 2.0f * (d[3] - d[1]) + (d[4] - d[2]);
    o[4] = 2.0f * (d[1] - d[3]) + (d[4] - d[2]);
    o[5] = (4.0f * d[1] - 5.0f * d[3]) + d[5];
}

// one line of A^T m
__device__ __forceinline__ void at6(const float4 (&m)[6], float4 (&o)[4])
{
    const float4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = (m[0] + s12) + s34;
    o[1] = d12 + 2.0f * d34;
    o[2] = s12 + 4.0f * s34;
    o[3] = (d12 + 8.0f * d34) + m[5];
}

struct WinoInArgs {
    WinoLevels lv;
    const float *x[IA_MAX_LEVELS];        // per level (B, H, W, Ctot) channels-last
    float *V;                             // (groups * 36, T, Cg)
    // optional pre-activation of the input: relu(x * scale[c] + shift[c]) -- the folded
    // BatchNorm + ReLU of the 1x1 convolution in front, applied on load instead of in a pass
    // of its own; padding stays zero
    const float *pre_scale, *pre_shift;
    int32_t Ctot, Cg, T, pre_relu, tpw;
};

// lane -> (tile, channel quad).  Layers with fewer than 256 channels put several tiles into one
// wavefront (64 channels: 4 tiles x 16 quads, 48 channels: 5 x 12) instead of leaving lanes idle:
// the 64- and 128-channel bottleneck layers ran at 2.5-3.8 TB/s with 16 / 32 live lanes, the
// 48-column reg|iou output transform at 0.7 TB/s with 12.
struct LaneMap { int t, c; bool on; };
// PACKED = false: one tile per wavefront -- the tile, its level and every table entry are
// wavefront-uniform (scalar registers, scalar loads)
template <bool PACKED>
__device__ __forceinline__ LaneMap lane_map(int Ctot, int T, int tpw)
{
    LaneMap m;
    const int lane = threadIdx.x;
    if (PACKED) {
        const int q = Ctot >> 2;                       // quads per tile, <= 32 here
        const int sub = lane / q;
        const int tw = xcd_tile(blockIdx.x, (T + tpw - 1) / tpw) * tpw + sub;
        m.t = tw; m.c = (lane - sub * q) * 4;
        m.on = (sub < tpw) && (tw < T);
    } else {
        m.t = __builtin_amdgcn_readfirstlane(xcd_tile(blockIdx.x, T));
        m.c = (blockIdx.y * 64 + lane) * 4;
        m.on = (m.c < Ctot) && (m.t < T);
    }
    return m;
}
static int tiles_per_wave(int channels)
{
    const int q = channels / 4;
    return (q < 64) ? (64 / q) : 1;
}

template <bool PACKED>
__global__ void __launch_bounds__(64) k_wino_in(WinoInArgs a)
{
    const LaneMap lm = lane_map<PACKED>(a.Ctot, a.T, a.tpw);
    if (!lm.on) return;
    const int t = lm.t, c = lm.c;
    int H, W;
    const TileRef r = locate_tile(a.lv, t, &H, &W);
    const float *x = level_ptr<!PACKED>(a.x, r.l) + (size_t)r.b * H * W * a.Ctot + c;
    const bool pre = a.pre_shift != nullptr;
    float4 ps = f4(1.0f), pb = f4(0.0f);
    if (pre) {
        if (a.pre_scale) ps = *reinterpret_cast<const float4 *>(a.pre_scale + c);
        pb = *reinterpret_cast<const float4 *>(a.pre_shift + c);
    }
    // All 36 loads are issued before anything consumes them.  (With the pre-activation's
    // `if (pre)` inside the load loop the compiler emitted load, branch, s_waitcnt vmcnt(0) 36
    // times over: ONE kilobyte in flight per wavefront at 3 wavefronts per SIMD.)
    float4 d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int y = r.y0 - 1 + i;
        const int yc = (y >= 0 && y < H) ? y : 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int xx = r.x0 - 1 + j;
            const int xc = (xx >= 0 && xx < W) ? xx : 0;
            // unconditional (clamped address); padding is selected afterwards
            d[i][j] = *reinterpret_cast<const float4 *>(x + ((size_t)yc * W + xc) * a.Ctot);
        }
    }
    if (pre) {
        const bool relu = a.pre_relu != 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float4 v = d[i][j];
                v = make_float4(v.x * ps.x + pb.x, v.y * ps.y + pb.y, v.z * ps.z + pb.z, v.w * ps.w + pb.w);
                const float4 z = make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f,
                                             v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f);
                d[i][j] = relu ? z : v;
            }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int y = r.y0 - 1 + i;
        const bool yin = (y >= 0) && (y < H);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int xx = r.x0 - 1 + j;
            const bool in = yin && (xx >= 0) && (xx < W);
            d[i][j] = in ? d[i][j] : f4(0.0f);
        }
    }
    float4 tmp[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {                       // columns: tmp = B^T d
        float4 col[6], o[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = d[i][j];
        bt6(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][j] = o[i];
    }
    const int g = c / a.Cg, cc = c - g * a.Cg;
    float *v = a.V + ((size_t)g * 36 * a.T + t) * a.Cg + cc;
    const size_t kstride = (size_t)a.T * a.Cg;
#pragma unroll
    for (int i = 0; i < 6; ++i) {                       // rows: V = tmp B
        float4 o[6];
        bt6(tmp[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j)
            *reinterpret_cast<float4 *>(v + (size_t)(i * 6 + j) * kstride) = o[j];
    }
}

// one line of A dy: the adjoint of at6 (4 values -> 6), A = (A^T)^T
__device__ __forceinline__ void a6(const float4 (&y)[4], float4 (&o)[6])
{
    const float4 s02 = y[0] + y[2], s13 = y[1] + y[3];
    const float4 t02 = y[0] + 4.0f * y[2], t13 = 2.0f * y[1] + 8.0f * y[3];
    o[0] = y[0];
    o[1] = s02 + s13;
    o[2] = s02 - s13;
    o[3] = t02 + t13;
    o[4] = t02 - t13;
    o[5] = y[3];
}

// Gradient of the output transform (training, weight gradient in the Winograd domain):
// dM = A dY A^T per tile, dY = the 4x4 output pixels of the tile (zero outside the feature map,
// where the forward output transform wrote nothing), scattered as 36 matrices like V.
template <bool PACKED>
__global__ void __launch_bounds__(64) k_wino_dy(WinoInArgs a)
{
    const LaneMap lm = lane_map<PACKED>(a.Ctot, a.T, a.tpw);
    if (!lm.on) return;
    const int t = lm.t, c = lm.c;
    int H, W;
    const TileRef r = locate_tile(a.lv, t, &H, &W);
    const float *x = level_ptr<!PACKED>(a.x, r.l) + (size_t)r.b * H * W * a.Ctot + c;
    float4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = r.y0 + i;
        const int yc = (y < H) ? y : (H - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = r.x0 + j;
            const int xc = (xx < W) ? xx : (W - 1);
            const float4 v = *reinterpret_cast<const float4 *>(x + ((size_t)yc * W + xc) * a.Ctot);
            d[i][j] = (y < H && xx < W) ? v : f4(0.0f);
        }
    }
    float4 tmp[6][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                       // columns: tmp = A d
        float4 col[4], o[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) col[i] = d[i][j];
        a6(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][j] = o[i];
    }
    const int g = c / a.Cg, cc = c - g * a.Cg;
    float *v = a.V + ((size_t)g * 36 * a.T + t) * a.Cg + cc;
    const size_t kstride = (size_t)a.T * a.Cg;
#pragma unroll
    for (int i = 0; i < 6; ++i) {                       // rows: dM = tmp A^T
        float4 o[6];
        a6(tmp[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j)
            *reinterpret_cast<float4 *>(v + (size_t)(i * 6 + j) * kstride) = o[j];
    }
}

struct WinoOutArgs {
    WinoLevels lv;
    const float *M;                       // (groups * 36, T, Cg)
    const float *bias;                    // (groups * Cg) or NULL
    WinoSeg seg[kMaxSeg];
    int32_t nseg, Ctot, Cg, T, relu, tpw;
};

template <bool PACKED>
__global__ void __launch_bounds__(64) k_wino_out(WinoOutArgs a)
{
    const LaneMap lm = lane_map<PACKED>(a.Ctot, a.T, a.tpw);
    if (!lm.on) return;
    const int t = lm.t, c = lm.c;
    int H, W;
    const TileRef r = locate_tile(a.lv, t, &H, &W);
    const int g = c / a.Cg, cc = c - g * a.Cg;
    const float *m = a.M + ((size_t)g * 36 * a.T + t) * a.Cg + cc;
    const size_t kstride = (size_t)a.T * a.Cg;
    // the bias is the OLDEST load: were it issued after the 36 loads of M, its s_waitcnt vmcnt(0)
    // would sit in the store phase, where (behind the per-pixel branches) it is repeated in front
    // of every pixel and waits for all the stores issued so far
    float4 bz = f4(0.0f);
    if (a.bias) bz = *reinterpret_cast<const float4 *>(a.bias + c);
    float4 s[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {                       // columns: s = A^T m
        float4 col[6], o[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = load_nt(m + (size_t)(i * 6 + j) * kstride);
        at6(col, o);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i][j] = o[i];
    }
    // Destination of this lane's four channels: the last segment that starts at or before c
    // (segments ascend).  Its parameters are selected ONCE into registers from scalar loads;
    // `a.seg[si]` with a per-lane si was a vector load from the kernarg segment that the compiler
    // re-issued in front of every one of the 16 stores (load, s_waitcnt vmcnt(0), store).
    int sc0 = opaque(a.seg[0].c0), sn = opaque(a.seg[0].n), sC = opaque(a.seg[0].Cdst), soff = opaque(a.seg[0].coff);
    float *sdst = level_ptr<!PACKED>(a.seg[0].dst, r.l);
#pragma unroll
    for (int k = 1; k < kMaxSeg; ++k) {
        const int kc0 = opaque(a.seg[k].c0), kn = opaque(a.seg[k].n), kC = opaque(a.seg[k].Cdst),
                  koff = opaque(a.seg[k].coff);
        float *kdst = level_ptr<!PACKED>(a.seg[k].dst, r.l);
        const bool m = (k < a.nseg) && (c >= kc0);
        sc0 = m ? kc0 : sc0; sn = m ? kn : sn; sC = m ? kC : sC; soff = m ? koff : soff;
        sdst = m ? kdst : sdst;
    }
    const bool whole = (c >= sc0) && (c + 4 <= sc0 + sn) && (((soff + c - sc0) & 3) == 0) && ((sC & 3) == 0);
    float *const dst0 = sdst + soff + (c - sc0);
    // a lane that straddles segments / padding, or whose destination is not 16-byte aligned
    // (the 9-channel IoU output): destination of each of its four channels, resolved once
    float *qdst[4] = {nullptr, nullptr, nullptr, nullptr};
    int qC[4] = {0, 0, 0, 0}, qoff[4] = {0, 0, 0, 0};
    if (!whole) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = c + q;
            for (int k = 0; k < a.nseg; ++k) {
                const WinoSeg &z = a.seg[k];
                if (ch >= z.c0 && ch < z.c0 + z.n) {
                    qdst[q] = level_ptr<!PACKED>(z.dst, r.l); qC[q] = z.Cdst; qoff[q] = z.coff + (ch - z.c0);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                       // rows: Y = s A
        float4 o[4];
        at6(s[i], o);
        const int y = r.y0 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = r.x0 + j;
            if (y >= H || xx >= W) continue;
            float4 v = o[j] + bz;
            if (a.relu) v = make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f,
                                        v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f);
            const size_t pix = ((size_t)r.b * H + y) * W + xx;
            if (whole) {
                *reinterpret_cast<float4 *>(dst0 + pix * sC) = v;
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (qdst[q]) qdst[q][pix * qC[q] + qoff[q]] = e[q];
            }
        }
    }
}

}  // namespace ia

extern "C" {

int ia_wino_tiles(const ia_wino_geom *g, int32_t *tiles)
{
    ia::WinoLevels w;
    int rc = ia::make_wino_levels(g, w);
    if (rc) return rc;
    if (tiles) *tiles = w.tile_off[w.L];
    return 0;
}

int ia_wino_input_transform(const ia_wino_geom *g, const float *const *x, int channels, int groups,
                            const float *pre_scale, const float *pre_shift, int pre_relu, float *V,
                            void *stream)
{
    ia::WinoInArgs a;
    int rc = ia::make_wino_levels(g, a.lv);
    if (rc) return rc;
    if (!x || !V || channels < 4 || (channels & 3) || groups < 1 || channels % groups) return IA_E_ARG;
    a.Ctot = channels; a.Cg = channels / groups; a.T = a.lv.tile_off[a.lv.L];
    if (a.Cg & 3) return IA_E_ARG;
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        a.x[l] = (l < a.lv.L) ? x[l] : nullptr;
        if (l < a.lv.L && (!x[l] || ((uintptr_t)x[l] & 15u))) return IA_E_ARG;
    }
    a.V = V;
    if (pre_scale && !pre_shift) return IA_E_ARG;
    a.pre_scale = pre_scale; a.pre_shift = pre_shift; a.pre_relu = pre_relu ? 1 : 0;
    a.tpw = ia::tiles_per_wave(channels);
    const int waves = (a.T + a.tpw - 1) / a.tpw;
    dim3 grid((unsigned)((waves + 7) / 8 * 8), (unsigned)((channels / 4 + 63) / 64));
    if (a.tpw > 1) hipLaunchKernelGGL(ia::k_wino_in<true>, grid, dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ia::k_wino_in<false>, grid, dim3(64), 0, (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}

int ia_wino_grad_output_transform(const ia_wino_geom *g, const float *const *dy, int channels,
                                   float *dM, void *stream)
{
    ia::WinoInArgs a;
    int rc = ia::make_wino_levels(g, a.lv);
    if (rc) return rc;
    if (!dy || !dM || channels < 4 || (channels & 3)) return IA_E_ARG;
    a.Ctot = channels; a.Cg = channels; a.T = a.lv.tile_off[a.lv.L];
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        a.x[l] = (l < a.lv.L) ? dy[l] : nullptr;
        if (l < a.lv.L && (!dy[l] || ((uintptr_t)dy[l] & 15u))) return IA_E_ARG;
    }
    a.V = dM;
    a.pre_scale = a.pre_shift = nullptr; a.pre_relu = 0;
    a.tpw = ia::tiles_per_wave(channels);
    const int waves = (a.T + a.tpw - 1) / a.tpw;
    dim3 grid((unsigned)((waves + 7) / 8 * 8), (unsigned)((channels / 4 + 63) / 64));
    if (a.tpw > 1) hipLaunchKernelGGL(ia::k_wino_dy<true>, grid, dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ia::k_wino_dy<false>, grid, dim3(64), 0, (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}

int ia_wino_output_transform(const ia_wino_geom *g, const float *M, int channels, int groups,
                             const float *bias, int relu, int nseg, const ia_wino_seg *segs,
                             void *stream)
{
    ia::WinoOutArgs a;
    int rc = ia::make_wino_levels(g, a.lv);
    if (rc) return rc;
    if (!M || channels < 4 || (channels & 3) || groups < 1 || channels % groups || !segs ||
        nseg < 1 || nseg > ia::kMaxSeg)
        return IA_E_ARG;
    a.M = M; a.bias = bias; a.Ctot = channels; a.Cg = channels / groups;
    if (a.Cg & 3) return IA_E_ARG;
    a.T = a.lv.tile_off[a.lv.L]; a.relu = relu ? 1 : 0; a.nseg = nseg;
    memset(a.seg, 0, sizeof(a.seg));
    for (int k = 0; k < nseg; ++k) {
        const ia_wino_seg &s = segs[k];
        if (s.c0 < 0 || s.n < 1 || s.c0 + s.n > channels || s.dst_channels < s.n + s.dst_offset ||
            s.dst_offset < 0)
            return IA_E_ARG;
        if (k > 0 && s.c0 < segs[k - 1].c0 + segs[k - 1].n) return IA_E_ARG;       // ascending
        a.seg[k].c0 = s.c0; a.seg[k].n = s.n; a.seg[k].Cdst = s.dst_channels; a.seg[k].coff = s.dst_offset;
        for (int l = 0; l < a.lv.L; ++l) {
            if (!s.dst[l]) return IA_E_ARG;
            a.seg[k].dst[l] = s.dst[l];
        }
    }
    a.tpw = ia::tiles_per_wave(channels);
    const int waves = (a.T + a.tpw - 1) / a.tpw;
    dim3 grid((unsigned)((waves + 7) / 8 * 8), (unsigned)((channels / 4 + 63) / 64));
    if (a.tpw > 1) hipLaunchKernelGGL(ia::k_wino_out<true>, grid, dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ia::k_wino_out<false>, grid, dim3(64), 0, (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}

}  // extern "C"
