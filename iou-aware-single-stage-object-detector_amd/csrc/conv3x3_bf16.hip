// 3x3 / stride 1 / pad 1 convolution on bf16 channels-last tensors, fp32 accumulation, bias (+ReLU)
// epilogue: the head-tower / FPN-output shape of BASELINE config 3 (R-101 bf16, reference
// iou_aware_retina_head.py:171-219 `ConvModule(256, 256, 3, padding=1)`, conv_module.py:149-163).
// An implicit GEMM on v_mfma_f32_32x32x16_bf16: M = the pixels of a spatial tile, N = 256 output
// channels, K = 9 taps x Cin.
//
//  * A workgroup (8 wavefronts, 2 x 4) owns a TH x TW pixel tile (TH * TW <= 256; the tile shape is
//    picked per feature map so that the tiles cover it with little overhang: 10 x 24 for 100 x 168,
//    9 x 28 for 50 x 84 ...) and 256 output channels; a wavefront 128 pixels x 64 channels
//    = 4 x 2 accumulator blocks of 32 x 32.
//  * K loop: Cin in chunks of 32.  The (TH + 2) x (TW + 2) halo patch of a chunk goes to LDS ONCE
//    and serves all nine taps as shifted windows (a tap only moves the wavefront's read offset):
//    1.3 reads of the activation instead of an im2col GEMM's 9.  The weights of (chunk, tap) --
//    256 x 32, packed contiguously by ia_conv3x3_bf16_pack -- are double-buffered in LDS; the next
//    tile's global loads are issued before the MFMAs of the current one.
//  * LDS rows (a pixel's / an output channel's 32 k-values = 64 bytes) are padded to 80 bytes: the
//    16 lanes of a ds_read_b128 group then start in 16 different 16-byte bank slots.
//  * Epilogue from the accumulators: + bias, ReLU, round to bf16; neighbouring lanes hold
//    neighbouring output channels of one pixel, so lane pairs swap one value (DPP) and store
//    4 bytes each, 64-byte runs per pixel.
#include <string.h>
#include <map>
#include <mutex>
#include <tuple>
#include "ia_internal.hpp"
#include "ia_math.hpp"

namespace ia {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kCvThreads = 512;
constexpr int kCvBK = 32;                 // input channels per K step
constexpr int kCvRow = 80;                // bytes per LDS row (64 of data)
constexpr int kCvMaxPatch = 352;          // (TH + 2) * (TW + 2) <= this (256-pixel tiles)
constexpr int kCvMaxPatchHalf = 208;      // ... for 128-pixel tiles
constexpr int kCvBN = 256;

constexpr int kCvMaxGroups = 2;

// One launch covers a list of feature maps (the pyramid levels of the shared-weight head) and up
// to two groups (the cls / reg towers: different inputs, weights and outputs, one tile list).
struct Conv3Args {
    const uint16_t *x[kCvMaxGroups][IA_MAX_LEVELS];   // (B, H_l, W_l, .) bf16, pixel stride xs
    uint16_t *y[kCvMaxGroups][IA_MAX_LEVELS];         // (B, H_l, W_l, .) bf16, pixel stride ys
    const uint16_t *wp;                   // packed weights [groups][ntile][Cin / 32][9][256][32]
    const float *bias;                    // (groups * Cout) or NULL
    int32_t L, B, Cin, Cout, xs, ys, relu, ntile;     // Cin / Cout per group; ntile = ceil(Cout / 256)
    int32_t H[IA_MAX_LEVELS], W[IA_MAX_LEVELS], TH[IA_MAX_LEVELS], TW[IA_MAX_LEVELS];
    int32_t tiles_y[IA_MAX_LEVELS], tiles_x[IA_MAX_LEVELS], tile_off[IA_MAX_LEVELS + 1];
};

__device__ __forceinline__ uint32_t bf16_rne(float f)
{
    uint32_t u = to_bits(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// MB = 32-pixel accumulator blocks per wavefront: 4 (256-pixel tiles) or 2 (128-pixel tiles, for
// feature maps whose 256-pixel tiles would leave most of the chip's workgroup slots empty)
template <int MB>
__global__ void __launch_bounds__(kCvThreads, 2) k_conv3x3_bf16(Conv3Args a)
{
    constexpr int kPatch = MB == 4 ? kCvMaxPatch : kCvMaxPatchHalf;
    constexpr int kWM = 32 * MB;                           // pixels per wavefront
    __shared__ __attribute__((aligned(16))) unsigned char s_a[kPatch * kCvRow];
    __shared__ __attribute__((aligned(16))) unsigned char s_b[2][kCvBN * kCvRow];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 2, wn = wv & 3;                  // 2 x 4 wavefronts
    int lv = 0;
    while (lv + 1 < a.L && (int)blockIdx.x >= a.tile_off[lv + 1]) ++lv;   // wavefront-uniform
    int t = (int)blockIdx.x - a.tile_off[lv];
    const int H = a.H[lv], W = a.W[lv], TH = a.TH[lv], TW = a.TW[lv];
    const int txi = t % a.tiles_x[lv]; t /= a.tiles_x[lv];
    const int tyi = t % a.tiles_y[lv];
    const int b = t / a.tiles_y[lv];
    const int grp = (int)blockIdx.y / a.ntile, nt = (int)blockIdx.y - grp * a.ntile;
    const int y0 = tyi * TH, x0 = txi * TW;
    const int PW = TW + 2, npix = (TH + 2) * PW, tile_px = TH * TW;
    const int nchunk = a.Cin / kCvBK, nsteps = nchunk * 9;

    // ---- this lane's four A rows (pixels) as byte offsets into the patch (tap (0,0) corner)
    int a_off[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int m = wm * kWM + mb * 32 + (lane & 31);
        m = m < tile_px ? m : tile_px - 1;                // idle rows read a valid address
        const int ty = m / TW, tx = m - ty * TW;
        a_off[mb] = (ty * PW + tx) * kCvRow + (lane >> 5) * 16;
    }
    const int b_off = (wn * 64 + (lane & 31)) * kCvRow + (lane >> 5) * 16;

    // ---- global -> register staging (plain scalars and macros: arrays captured by a lambda went
    // to scratch memory).  Weights of step s + 2 are requested at the start of step s and stored to
    // LDS at the end of step s + 1: two steps of MFMA time to arrive; the halo patch of the next
    // chunk is requested three taps ahead.
    const uint16_t *xb = a.x[grp][lv] + (size_t)b * H * W * a.xs;
    const uint16_t *wbase = a.wp + (size_t)(grp * a.ntile + nt) * nchunk * 9 * (kCvBN * kCvBK) + tid * 8;
    // patch pieces of this thread: p = u * 512 + tid -> pixel p >> 2, 16-byte part p & 3
    const uint16_t *pa0, *pa1, *pa2;
    bool in0, in1, in2, on0, on1, on2;
#define CV_PIECE(u, PA, IN, ON)                                                                     \
    {                                                                                               \
        int p = u * kCvThreads + tid;                                                               \
        ON = p < npix * 4;                                                                          \
        p = ON ? p : 0;                                                                             \
        const int px = p >> 2, part = p & 3;                                                        \
        const int py = px / PW, pxx = px - py * PW;                                                 \
        const int iy = y0 + py - 1, ix = x0 + pxx - 1;                                              \
        IN = ON && iy >= 0 && iy < H && ix >= 0 && ix < W;                                      \
        const int cy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix); \
        PA = xb + ((size_t)cy * W + cx) * a.xs + part * 8;                                       \
    }
    CV_PIECE(0, pa0, in0, on0)
    CV_PIECE(1, pa1, in1, on1)
    if (MB == 4) CV_PIECE(2, pa2, in2, on2) else { pa2 = pa1; in2 = on2 = false; }
#undef CV_PIECE
    // (an AND with a per-piece mask: `in ? value : zero` on a 128-bit value became a table in scratch)
    const uint32_t mk0 = in0 ? 0xffffffffu : 0u, mk1 = in1 ? 0xffffffffu : 0u, mk2 = in2 ? 0xffffffffu : 0u;
    unsigned char *sa0 = s_a + (tid >> 2) * kCvRow + (tid & 3) * 16;
    unsigned char *sa1 = sa0 + (kCvThreads >> 2) * kCvRow, *sa2 = sa1 + (kCvThreads >> 2) * kCvRow;
    const int sb_off = (tid >> 2) * kCvRow + (tid & 3) * 16;               // + 128 rows for the second piece
    uint4 ra0, ra1, ra2 = make_uint4(0u, 0u, 0u, 0u), rx0, rx1, ry0, ry1;
#define CV_LOAD_A(chunk)                                                                            \
    {                                                                                               \
        ra0 = *reinterpret_cast<const uint4 *>(pa0 + (chunk) * kCvBK);                              \
        ra1 = *reinterpret_cast<const uint4 *>(pa1 + (chunk) * kCvBK);                              \
        if (MB == 4) ra2 = *reinterpret_cast<const uint4 *>(pa2 + (chunk) * kCvBK);                 \
    }
#define CV_STORE_A()                                                                                \
    {                                                                                               \
        /* the padding select happens HERE: at the load it would be the load's first use, i.e. a  \
           full memory round trip in the middle of the K loop */                                    \
        if (on0) *reinterpret_cast<uint4 *>(sa0) = make_uint4(ra0.x & mk0, ra0.y & mk0, ra0.z & mk0, ra0.w & mk0); \
        if (on1) *reinterpret_cast<uint4 *>(sa1) = make_uint4(ra1.x & mk1, ra1.y & mk1, ra1.z & mk1, ra1.w & mk1); \
        if (MB == 4 && on2) *reinterpret_cast<uint4 *>(sa2) = make_uint4(ra2.x & mk2, ra2.y & mk2, ra2.z & mk2, ra2.w & mk2); \
    }
#define CV_LOAD_B(step, R0, R1)                                                                     \
    {                                                                                               \
        const uint16_t *src = wbase + (size_t)(step) * (kCvBN * kCvBK);                             \
        R0 = *reinterpret_cast<const uint4 *>(src);                                                 \
        R1 = *reinterpret_cast<const uint4 *>(src + kCvThreads * 8);                                \
    }
#define CV_STORE_B(buf, R0, R1)                                                                     \
    {                                                                                               \
        *reinterpret_cast<uint4 *>(s_b[buf] + sb_off) = R0;                                         \
        *reinterpret_cast<uint4 *>(s_b[buf] + sb_off + 128 * kCvRow) = R1;                          \
    }

    f32x16 acc[MB][2];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    CV_LOAD_A(0)
    CV_LOAD_B(0, rx0, rx1)
    CV_STORE_A()
    CV_STORE_B(0, rx0, rx1)
    CV_LOAD_B(nsteps > 1 ? 1 : 0, rx0, rx1)                // set X holds B(s + 1) at the start of step s
    __syncthreads();

    // One step: MFMAs of (chunk, TAP) from the patch and s_b[buf]; set X (weights of step + 1) goes
    // to s_b[buf ^ 1] at the end, set Y receives the weights of step + 2.  Branch-free on purpose:
    // every load / LDS store is unconditional (indices clamped at the end of the K loop), so the
    // compiler can count the outstanding loads and wait for exactly the older set -- behind a
    // conditional load it falls back to s_waitcnt vmcnt(0), which turns the two-step prefetch
    // into none.
#define CV_STEP(TAP, X0, X1, Y0, Y1)                                                                \
    {                                                                                               \
        constexpr int dy = (TAP) / 3, dx = (TAP) - dy * 3;                                          \
        const int step = step0 + (TAP);                                                             \
        const int s2 = step + 2 < nsteps ? step + 2 : nsteps - 1;                                   \
        CV_LOAD_B(s2, Y0, Y1)                                                                       \
        if ((TAP) == 5) CV_LOAD_A(chunk + 1 < nchunk ? chunk + 1 : chunk)                           \
        const unsigned char *ab = s_a + (dy * PW + dx) * kCvRow;                                    \
        const unsigned char *bb = s_b[0] + ((step & 1) ? kCvBN * kCvRow : 0) + b_off;               \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                          \
            bf16x8 fa[MB], fb[2];                                                                   \
            _Pragma("unroll") for (int i = 0; i < MB; ++i)                                          \
                fa[i] = *reinterpret_cast<const bf16x8 *>(ab + a_off[i] + kk * 32);                 \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
                fb[j] = *reinterpret_cast<const bf16x8 *>(bb + j * 32 * kCvRow + kk * 32);          \
            _Pragma("unroll") for (int i = 0; i < MB; ++i)                                          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                       \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0); \
        }                                                                                           \
        {                                                                                           \
            unsigned char *dst = s_b[0] + ((step & 1) ? 0 : kCvBN * kCvRow) + sb_off;               \
            *reinterpret_cast<uint4 *>(dst) = X0;                                                   \
            *reinterpret_cast<uint4 *>(dst + 128 * kCvRow) = X1;                                    \
        }                                                                                           \
        if ((TAP) == 8) {                                                                           \
            __syncthreads();                               /* every wavefront is done with the patch */ \
            CV_STORE_A()                                                                            \
        }                                                                                           \
        __syncthreads();                                                                            \
    }
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int step0 = chunk * 9;
        CV_STEP(0, rx0, rx1, ry0, ry1)
        CV_STEP(1, ry0, ry1, rx0, rx1)
        CV_STEP(2, rx0, rx1, ry0, ry1)
        CV_STEP(3, ry0, ry1, rx0, rx1)
        CV_STEP(4, rx0, rx1, ry0, ry1)
        CV_STEP(5, ry0, ry1, rx0, rx1)
        CV_STEP(6, rx0, rx1, ry0, ry1)
        CV_STEP(7, ry0, ry1, rx0, rx1)
        CV_STEP(8, rx0, rx1, ry0, ry1)
        // nine steps: the sets have swapped roles
        { const uint4 t0 = rx0, t1 = rx1; rx0 = ry0; rx1 = ry1; ry0 = t0; ry1 = t1; }
    }
#undef CV_STEP
#undef CV_LOAD_A
#undef CV_STORE_A
#undef CV_LOAD_B
#undef CV_STORE_B

    // ---- epilogue: C block (i, j): column n = lane & 31, row m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int odd = lane & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nt * kCvBN + wn * 64 + j * 32 + (lane & 31);          // channel inside the group
        const bool n_ok = (n - odd) + 1 < a.Cout;                             // the pair this lane stores (Cout is even)
        const float bz = (a.bias && n < a.Cout) ? a.bias[grp * a.Cout + n] : 0.0f;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float v0 = acc[i][j][r] + bz, v1 = acc[i][j][r + 1] + bz;
                if (a.relu) { v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; }
                // even lane keeps row r and takes the odd neighbour's row-r value (channel n + 1);
                // odd lane keeps row r + 1 and takes the even neighbour's (channel n - 1)
                const float give = odd ? v0 : v1;
                const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
                const int rr = odd ? r + 1 : r;
                const int m = wm * kWM + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
                const uint32_t lo = bf16_rne(odd ? got : v0), hi = bf16_rne(odd ? v1 : got);
                if (m < tile_px) {
                    const int ty = m / TW, tx = m - ty * TW;
                    const int oy = y0 + ty, ox = x0 + tx;
                    if (oy < H && ox < W && n_ok)
                        *reinterpret_cast<uint32_t *>(a.y[grp][lv] + (((size_t)b * H + oy) * W + ox) * a.ys + (n - odd)) = lo | (hi << 16);
                }
            }
        }
    }
}

// weights (groups * Cout, 3, 3, Cin) bf16 -> [groups][ceil(Cout / 256)][Cin / 32][9][256][32], output
// channels beyond Cout zero
__global__ void __launch_bounds__(256) k_conv3x3_pack(const uint16_t *w, uint16_t *wp, int Cin, int Cout, int groups)
{
    const int ntile = (Cout + kCvBN - 1) / kCvBN, nchunk = Cin / kCvBK;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)groups * ntile * kCvBN * 9 * Cin;
    if (idx >= total) return;
    const int k = (int)(idx % kCvBK);
    int64_t r = idx / kCvBK;
    const int n = (int)(r % kCvBN); r /= kCvBN;
    const int tap = (int)(r % 9); r /= 9;
    const int chunk = (int)(r % nchunk); r /= nchunk;
    const int nt = (int)(r % ntile);
    const int g = (int)(r / ntile);
    const int co = nt * kCvBN + n;
    wp[idx] = co < Cout ? w[(((size_t)(g * Cout + co) * 9) + tap) * Cin + chunk * kCvBK + k] : (uint16_t)0;
}

// tile shape for an H x W map: TH * TW <= max_px, patch <= max_patch, least overhang
static void conv3_tile_shape_search(int H, int W, int max_px, int max_patch, int &TH, int &TW);

// memoised per (H, W, tile size): the search walks ~3 800 shapes, twice per level and launch
static void conv3_tile_shape(int H, int W, int max_px, int max_patch, int &TH, int &TW)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, std::pair<int, int>> memo;
    const auto key = std::make_tuple(H, W, max_px, max_patch);
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = memo.find(key);
        if (it != memo.end()) { TH = it->second.first; TW = it->second.second; return; }
    }
    conv3_tile_shape_search(H, W, max_px, max_patch, TH, TW);
    std::lock_guard<std::mutex> lock(mu);
    if (memo.size() > 4096) memo.clear();
    memo[key] = std::make_pair(TH, TW);
}

static void conv3_tile_shape_search(int H, int W, int max_px, int max_patch, int &TH, int &TW)
{
    double best = -1.0;
    TH = 8; TW = 16;
    for (int tw = 4; tw <= 64; ++tw) {
        for (int th = 2; th <= 64; ++th) {
            if (th * tw > max_px || (th + 2) * (tw + 2) > max_patch) continue;
            const int64_t ty = (H + th - 1) / th, tx = (W + tw - 1) / tw;
            const double eff = (double)H * W / ((double)ty * tx * max_px);      // useful rows per tile
            if (eff > best + 1e-9) { best = eff; TH = th; TW = tw; }
        }
    }
}

}  // namespace ia

extern "C" {

size_t ia_conv3x3_bf16_packed_bytes(int Cin, int Cout, int groups)
{
    return (size_t)groups * ((Cout + ia::kCvBN - 1) / ia::kCvBN) * ia::kCvBN * 9 * Cin * 2;
}

int ia_conv3x3_bf16_pack(const void *w, int Cin, int Cout, int groups, void *wp, void *stream)
{
    if (!w || !wp || Cin < 32 || (Cin % ia::kCvBK) || Cout < 2 || (Cout & 1) || groups < 1 ||
        groups > ia::kCvMaxGroups)
        return IA_E_ARG;
    const int64_t total = (int64_t)(ia_conv3x3_bf16_packed_bytes(Cin, Cout, groups) / 2);
    hipLaunchKernelGGL(ia::k_conv3x3_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const uint16_t *>(w), static_cast<uint16_t *>(wp), Cin, Cout, groups);
    return ia::hip_status(hipGetLastError());
}

int ia_conv3x3_bf16_levels(const ia_conv3x3_desc *d, const void *wp, const float *bias, int relu, void *stream)
{
    if (!d || !wp || d->num_levels < 1 || d->num_levels > IA_MAX_LEVELS || d->batch < 1 || d->groups < 1 ||
        d->groups > ia::kCvMaxGroups || d->cin < 32 || (d->cin % ia::kCvBK) || d->cout < 2 || (d->cout & 1) ||
        d->x_stride < d->cin || (d->x_stride & 7) || d->y_stride < d->cout || (d->y_stride & 1))
        return IA_E_ARG;
    if ((uintptr_t)wp & 15u) return IA_E_ARG;
    ia::Conv3Args a;
    memset(&a, 0, sizeof(a));
    a.wp = static_cast<const uint16_t *>(wp); a.bias = bias;
    a.L = d->num_levels; a.B = d->batch; a.Cin = d->cin; a.Cout = d->cout; a.xs = d->x_stride; a.ys = d->y_stride;
    a.relu = relu ? 1 : 0; a.ntile = (d->cout + ia::kCvBN - 1) / ia::kCvBN;
    for (int l = 0; l < d->num_levels; ++l) {
        if (d->H[l] < 1 || d->W[l] < 1) return IA_E_ARG;
        a.H[l] = d->H[l]; a.W[l] = d->W[l];
        for (int g = 0; g < d->groups; ++g) {
            if (!d->x[g][l] || !d->y[g][l] || ((uintptr_t)d->x[g][l] & 15u) || ((uintptr_t)d->y[g][l] & 3u))
                return IA_E_ARG;
            a.x[g][l] = static_cast<const uint16_t *>(d->x[g][l]);
            a.y[g][l] = static_cast<uint16_t *>(d->y[g][l]);
        }
    }
    // 256-pixel tiles unless they fill less than a round and a half of the chip's 512 resident
    // workgroups: then 128-pixel tiles (twice the workgroups, half the accumulators each)
    int64_t tiles = 0;
    int mb = 4;
    for (int pass = 0; pass < 2; ++pass) {
        tiles = 0;
        for (int l = 0; l < d->num_levels; ++l) {
            ia::conv3_tile_shape(a.H[l], a.W[l], 64 * mb, mb == 4 ? ia::kCvMaxPatch : ia::kCvMaxPatchHalf, a.TH[l], a.TW[l]);
            a.tiles_y[l] = (a.H[l] + a.TH[l] - 1) / a.TH[l]; a.tiles_x[l] = (a.W[l] + a.TW[l] - 1) / a.TW[l];
            a.tile_off[l] = (int32_t)tiles;
            tiles += (int64_t)d->batch * a.tiles_y[l] * a.tiles_x[l];
            if (tiles > 2147483647LL) return IA_E_ARG;
        }
        if (mb == 2 || tiles * d->groups * a.ntile >= 768) break;
        mb = 2;
    }
    for (int l = d->num_levels; l <= IA_MAX_LEVELS; ++l) a.tile_off[l] = (int32_t)tiles;
    const dim3 grid((unsigned)tiles, (unsigned)(d->groups * a.ntile));
    if (mb == 4) hipLaunchKernelGGL(ia::k_conv3x3_bf16<4>, grid, dim3(ia::kCvThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ia::k_conv3x3_bf16<2>, grid, dim3(ia::kCvThreads), 0, (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}

}  // extern "C"
