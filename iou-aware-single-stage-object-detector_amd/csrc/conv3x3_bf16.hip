// 3x3 / stride 1 / pad 1 convolution on bf16 channels-last tensors, fp32 accumulation, bias (+ReLU)
// epilogue: the head-tower / FPN-output / bottleneck shape of BASELINE config 3 (R-101 bf16, reference
// iou_aware_retina_head.py:171-219 `ConvModule(256, 256, 3, padding=1)`, conv_module.py:149-163,
// resnet.py:215-255).  An implicit GEMM on v_mfma_f32_32x32x16_bf16: M = the pixels of a spatial tile,
// N = up to 256 output channels, K = 9 taps x Cin.
//
//  * A workgroup = four wavefronts as WM x WN, each 32 * MB pixels x 64 output channels (MB x 2
//    accumulator blocks of 32 x 32); it owns a TH x TW pixel tile (64 or 128 pixels; the shape is picked
//    per feature map for the least overhang) and 64 * WN channels.  Two or three workgroups share a CU,
//    so one's prologue and epilogue run under the others' K loops.
//  * K loop: Cin in chunks of 32.  The (TH + 2) x (TW + 2) halo patch of a chunk goes to LDS ONCE
//    (double-buffered: requested at tap 2, stored at tap 6) and serves all nine taps as shifted windows:
//    1.3 reads of the activation instead of an im2col GEMM's 9.
//  * The weights never touch LDS: ia_conv3x3_bf16_pack stores them in FRAGMENT order, a wavefront's four
//    B fragments of a step are four 1-KiB-contiguous loads straight into registers (three register
//    sets in rotation, requested two steps ahead).
//  * LDS rows (a pixel's 32 k-values = 64 bytes) are padded to 80 bytes and the patch's row pitch is
//    TW + 16 pixels: the 16 lanes of every ds_read_b128 service group start in 16 different bank slots.
//  * PIPE = 2 (every variant with two or four pixel blocks per wavefront): the A fragments run through a ring of four registers, read
//    three MFMA pairs ahead of their use.
//  * Epilogue (ia_conv3.hpp): + bias, ReLU, one rounding to bf16 (v_cvt_pk_bf16_f32); lane pairs swap one
//    value (DPP) and store 4 bytes each, 64-byte runs per pixel; rows advance by additions.
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <tuple>
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_conv3.hpp"

namespace ia {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kCvBK = 32;                 // input channels per K step
constexpr int kCvRow = 80;                // bytes per LDS row (64 of data)
// The halo patch of a TH x TW tile lies in LDS with a row pitch of TW + 16 pixels (not TW + 2): pixel
// m = ty * TW + tx of the tile then sits in LDS row R = m + 16 ty + (tap offset), R = m (mod 16), and the
// 16 lanes of every ds_read_b128 service group (lanes {0-3, 12-15, 20-27} ...: rows distinct mod 16)
// fall into 16 different 16-byte bank slots whatever the tile shape.  With the dense pitch TW + 2 every
// tile row boundary inside a 32-pixel block made two or three lanes of a group collide -- each
// collision a whole extra LDS cycle for the group (8 x 8 tiles: 3 cycles instead of 1).
constexpr int kCvPitchPad = 16;
constexpr int kCvMaxRows128 = 320;        // (TH + 2) * (TW + 16) <= this for 128-pixel tiles (8 x 16, 4 x 32)
constexpr int kCvMaxRows64 = 240;         // ... for 64-pixel tiles (8 x 8, 4 x 16, 2 x 32)
constexpr int kCvMaxPatchPx128 = 256;     // (TH + 2) * (TW + 2) <= this: four 16-byte pieces per thread
constexpr int kCvMaxPatchPx64 = 128;      // ... two pieces per thread (8 x 8, 4 x 16)
constexpr int kCvBigTiles = 2048;         // maps with this many 128-pixel tiles in the batch take (4, 1, 4)
constexpr int kCvBN = 256;

__device__ __forceinline__ uint32_t bf16_rne(float f)
{
    uint32_t u = to_bits(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// MB = 32-pixel accumulator blocks per wavefront: 4 (256-pixel tiles) or 2 (128-pixel tiles, for
// feature maps whose 256-pixel tiles would leave most of the chip's workgroup slots empty)
// WM x WN wavefronts, each 32 * MB pixels x 64 output channels: a tile of 32 * MB * WM pixels and
// 64 * WN channels.  Four wavefronts per workgroup: at 160-250 registers a CU holds 8-12 wavefronts,
// i.e. two or three workgroups, and one's prologue / epilogue overlaps the others' K loops (the
// 8-wavefront workgroup of round 3, alone on its CU, ran 12-40 % slower).  WN = 4 for Cout > 128;
// the narrow backbone layers turn column wavefronts into row wavefronts (WN = 2 / 1).
// PIPE = 2 (the (4, 1, 4) variant): the A fragments software-pipelined through a ring of four registers
// -- the fragment of MFMA pair p + 3 is requested in front of pair p (a step = 8 pairs of MFMAs on one
// A fragment; pairs 8..10 = the next step's first three), so no LDS latency stands in front of any
// MFMA: `s_waitcnt lgkmcnt(3)` instead of the old `ds_read x2; lgkmcnt(1); v_mfma` at the head of
// every step.  Needs the next chunk's patch one step early (barrier behind tap 7) and a second
// barrier per chunk (behind tap 2) between the last reads of a patch buffer and the stores that
// overwrite it.  16 registers for the ring (the half-step form -- the kk = 1 fragments under the kk = 0
// MFMAs, the next step's kk = 0 fragments under the kk = 1 MFMAs, 32 registers -- spilled at the 256
// the two-wavefront-per-SIMD budget allows and lost: 0.322 against 0.312 ms).  Round 5, 100 x 168 at
// batch 16: 0.312 -> 0.300 ms; counters: matrix pipe 0.53 -> 0.59 busy, and the chip answers with 1.67
// instead of 1.80 GHz (profiles/r05_conv3x3_bf16_pmc.txt): busy x clock, i.e. throughput, +2.4 %.
template <int MB, int WM, int WN, int PIPE = 0>
// (waves per SIMD asked for: 2 for the 128 / 256-accumulator variants; the (1, 4, 1) variant's two 25 KiB
// patch buffers allow three workgroups per CU, (1, 2, 2) with its 64-pixel patches four)
__global__ void __launch_bounds__(64 * WM * WN, MB == 1 ? (WM == 4 ? 3 : 4) : 2) k_conv3x3_bf16(Conv3Args a)
{
    constexpr int kThreads = 64 * WM * WN;
    static_assert(MB * WM == 4 || MB * WM == 2, "128- or 64-pixel tiles");
    constexpr int kPatch = MB * WM == 4 ? kCvMaxRows128 : kCvMaxRows64;          // LDS rows of one patch buffer
    constexpr int NP = ((MB * WM == 4 ? kCvMaxPatchPx128 : kCvMaxPatchPx64) * 4 + kThreads - 1) / kThreads;   // patch pieces per thread
    static_assert(NP <= 4, "patch pieces");
    constexpr int kWM = 32 * MB;                           // pixels per wavefront
    __shared__ __attribute__((aligned(16))) unsigned char s_a[2][kPatch * kCvRow];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    int lv = 0;
    while (lv + 1 < a.L && (int)blockIdx.x >= a.tile_off[lv + 1]) ++lv;   // wavefront-uniform
    int t = (int)blockIdx.x - a.tile_off[lv];
    const int H = a.H[lv], W = a.W[lv], TH = a.TH[lv], TW = a.TW[lv];
    const int txi = t % a.tiles_x[lv]; t /= a.tiles_x[lv];
    const int tyi = t % a.tiles_y[lv];
    const int b = t / a.tiles_y[lv];
    const int grp = (int)blockIdx.y / a.ntile, nt = (int)blockIdx.y - grp * a.ntile;
    const int y0 = tyi * TH, x0 = txi * TW;
    const int PW = TW + 2, PWl = TW + kCvPitchPad, npix = (TH + 2) * PW, tile_px = TH * TW;
    const int nchunk = a.Cin / kCvBK, nsteps = nchunk * 9;

    // ---- this lane's A rows (pixels) as byte offsets into the patch (tap (0,0) corner)
    // (one vector division per lane; the blocks 32 pixels further on by additions and a wrap test --
    // 32 = q32 * TW + r32 with scalar q32, r32: the prologue's dozen vector integer divisions were
    // ~400 instructions of every workgroup's life)
    int a_off[MB];
    {
        const int q32 = 32 / TW, r32 = 32 - q32 * TW;
        const int m0 = wm * kWM + (lane & 31);
        int ty = m0 / TW, tx = m0 - ty * TW;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const bool idle = m0 + mb * 32 >= tile_px;    // idle rows read a valid address: the tile's last pixel
            const int ry = idle ? TH - 1 : ty, rx = idle ? TW - 1 : tx;
            a_off[mb] = (ry * PWl + rx) * kCvRow + (lane >> 5) * 16;
            tx += r32; ty += q32;
            if (tx >= TW) { tx -= TW; ++ty; }
        }
    }

    // ---- global -> register staging (plain scalars and macros: arrays captured by a lambda went
    // to scratch memory).  The weight fragments of step s + 2 are requested at the start of step s
    // (three register sets in rotation); the halo patch of the next chunk is requested at tap 2 and
    // stored to the other LDS buffer at tap 6.
    const uint16_t *xb = a.x[grp][lv] + (size_t)b * H * W * a.xs;
    // packed weights: [group][ntile][step][wn][j][kk][lane][8]: a wavefront's fragment load is 1 KiB contiguous
    const uint16_t *wbase = a.wp + (size_t)(grp * a.ntile + nt) * nsteps * (kCvBN * kCvBK) + (wn * 4 * 64 + lane) * 8;
    // patch pieces of this thread: p = u * kThreads + tid -> pixel p >> 2, 16-byte part p & 3
    const uint16_t *pa0, *pa1, *pa2, *pa3;
    int sa_off0, sa_off1, sa_off2, sa_off3;                // LDS byte offsets of the pieces
    bool in0, in1, in2, in3, on0, on1, on2, on3;
    // piece u of a thread = patch pixel (tid >> 2) + u * (kThreads / 4): row / column by additions from
    // piece 0's (one vector division)
    const int pq = (kThreads / 4) / PW, pr = (kThreads / 4) - pq * PW;
    int ppy = (tid >> 2) / PW, ppx = (tid >> 2) - ppy * PW;
#define CV_PIECE(u, PA, IN, ON, SA)                                                                 \
    {                                                                                               \
        const int p = u * kThreads + tid;                                                           \
        ON = p < npix * 4;                                                                          \
        const int part = p & 3;                                                                     \
        const int py = ON ? ppy : 0, pxx = ON ? ppx : 0;                                            \
        ppx += pr; ppy += pq;                                                                       \
        if (ppx >= PW) { ppx -= PW; ++ppy; }                                                        \
        const int iy = y0 + py - 1, ix = x0 + pxx - 1;                                              \
        IN = ON && iy >= 0 && iy < H && ix >= 0 && ix < W;                                      \
        const int cy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix); \
        PA = xb + ((size_t)cy * W + cx) * a.xs + part * 8;                                       \
        SA = (py * PWl + pxx) * kCvRow + part * 16;                                                 \
    }
    CV_PIECE(0, pa0, in0, on0, sa_off0)
    CV_PIECE(1, pa1, in1, on1, sa_off1)
    if (NP > 2) CV_PIECE(2, pa2, in2, on2, sa_off2) else { pa2 = pa1; in2 = on2 = false; sa_off2 = 0; }
    if (NP > 3) CV_PIECE(3, pa3, in3, on3, sa_off3) else { pa3 = pa1; in3 = on3 = false; sa_off3 = 0; }
#undef CV_PIECE
    // (an AND with a per-piece mask: `in ? value : zero` on a 128-bit value became a table in scratch)
    const uint32_t mk0 = in0 ? 0xffffffffu : 0u, mk1 = in1 ? 0xffffffffu : 0u, mk2 = in2 ? 0xffffffffu : 0u,
                   mk3 = in3 ? 0xffffffffu : 0u;
    uint4 ra0, ra1, ra2 = make_uint4(0u, 0u, 0u, 0u), ra3 = make_uint4(0u, 0u, 0u, 0u);
    uint4 bx0, bx1, bx2, bx3, by0, by1, by2, by3, bz0, bz1, bz2, bz3;     // weight fragments (j, kk) = (0,0) (0,1) (1,0) (1,1)
#define CV_LOAD_A(chunk)                                                                            \
    {                                                                                               \
        ra0 = *reinterpret_cast<const uint4 *>(pa0 + (chunk) * kCvBK);                              \
        ra1 = *reinterpret_cast<const uint4 *>(pa1 + (chunk) * kCvBK);                              \
        if (NP > 2) ra2 = *reinterpret_cast<const uint4 *>(pa2 + (chunk) * kCvBK);                  \
        if (NP > 3) ra3 = *reinterpret_cast<const uint4 *>(pa3 + (chunk) * kCvBK);                  \
    }
#define CV_STORE_A(buf)                                                                             \
    {                                                                                               \
        /* the padding select happens HERE: at the load it would be the load's first use, i.e. a  \
           full memory round trip in the middle of the K loop */                                    \
        unsigned char *dstA = s_a[0] + (buf) * (kPatch * kCvRow);                                   \
        if (on0) *reinterpret_cast<uint4 *>(dstA + sa_off0) = make_uint4(ra0.x & mk0, ra0.y & mk0, ra0.z & mk0, ra0.w & mk0); \
        if (on1) *reinterpret_cast<uint4 *>(dstA + sa_off1) = make_uint4(ra1.x & mk1, ra1.y & mk1, ra1.z & mk1, ra1.w & mk1); \
        if (NP > 2 && on2) *reinterpret_cast<uint4 *>(dstA + sa_off2) = make_uint4(ra2.x & mk2, ra2.y & mk2, ra2.z & mk2, ra2.w & mk2); \
        if (NP > 3 && on3) *reinterpret_cast<uint4 *>(dstA + sa_off3) = make_uint4(ra3.x & mk3, ra3.y & mk3, ra3.z & mk3, ra3.w & mk3); \
    }
#define CV_LOAD_B(step, R0, R1, R2, R3)                                                             \
    {                                                                                               \
        const uint16_t *src = wbase + (size_t)(step) * (kCvBN * kCvBK);                             \
        R0 = *reinterpret_cast<const uint4 *>(src);                                                 \
        R1 = *reinterpret_cast<const uint4 *>(src + 64 * 8);                                        \
        R2 = *reinterpret_cast<const uint4 *>(src + 2 * 64 * 8);                                    \
        R3 = *reinterpret_cast<const uint4 *>(src + 3 * 64 * 8);                                    \
    }

    f32x16 acc[MB][2];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    CV_LOAD_A(0)
    CV_LOAD_B(0, bx0, bx1, bx2, bx3)
    CV_LOAD_B(nsteps > 1 ? 1 : 0, by0, by1, by2, by3)
    CV_STORE_A(0)
    __syncthreads();
    bf16x8 fr0, fr1, fr2, fr3;                // PIPE: ring of four fragments, read three MFMA pairs ahead
    if constexpr (PIPE == 2) {
        static_assert(PIPE != 2 || MB == 4 || MB == 2, "the fragment ring is laid out for four or two pixel blocks");
        fr0 = *reinterpret_cast<const bf16x8 *>(s_a[0] + a_off[0]);
        fr1 = *reinterpret_cast<const bf16x8 *>(s_a[0] + a_off[1 % MB]);
        fr2 = *reinterpret_cast<const bf16x8 *>(s_a[0] + a_off[MB == 4 ? 2 : 0] + (MB == 4 ? 0 : 32));
    }

    // One step: the MFMAs of (chunk, TAP): A from the patch (a tap only shifts the read window), B
    // from the register set X; set Z receives the fragments of step + 2.  No barrier inside a
    // chunk: the wavefronts drift apart and one's LDS reads overlap the others' MFMAs.  Branch-free
    // on purpose (indices clamped at the end of the K loop): behind a conditional load the compiler
    // can no longer count the outstanding loads and waits vmcnt(0).
#define CV_MFMA(KK, F0, F1)                                                                         \
    {                                                                                               \
        bf16x8 fa[MB];                                                                              \
        _Pragma("unroll") for (int i = 0; i < MB; ++i)                                              \
            fa[i] = *reinterpret_cast<const bf16x8 *>(ab + a_off[i] + (KK) * 32);                   \
        const bf16x8 f0 = __builtin_bit_cast(bf16x8, F0), f1 = __builtin_bit_cast(bf16x8, F1);      \
        _Pragma("unroll") for (int i = 0; i < MB; ++i) {                                            \
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], f0, acc[i][0], 0, 0, 0);     \
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], f1, acc[i][1], 0, 0, 0);     \
        }                                                                                           \
    }
#define CV_PAIR(DST, SRC, USE, I, G0, G1)                                                          \
    {                                                                                               \
        DST = *reinterpret_cast<const bf16x8 *>(SRC);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        acc[I][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(USE, G0, acc[I][0], 0, 0, 0);           \
        acc[I][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(USE, G1, acc[I][1], 0, 0, 0);           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
#define CV_STEP(TAP, X0, X1, X2, X3, Z0, Z1, Z2, Z3)                                                \
    {                                                                                               \
        constexpr int dy = (TAP) / 3, dx = (TAP) - dy * 3;                                          \
        const int step = step0 + (TAP);                                                             \
        const int s2 = step + 2 < nsteps ? step + 2 : nsteps - 1;                                   \
        if ((TAP) == 2) CV_LOAD_A(chunk + 1 < nchunk ? chunk + 1 : chunk)                           \
        CV_LOAD_B(s2, Z0, Z1, Z2, Z3)                                                               \
        /* the loads stay HERE: left alone, the scheduler sinks them to just before their use two \
           steps later (one register set, a memory round trip per step).  The A fragments are    \
           left to it: each pair read right in front of its two MFMAs measured FASTER than all    \
           eight requested at the head of the step (0.363 against 0.396 ms on 100 x 168), and    \
           than a hand-pipelined order -- kk-1 fragments under the kk-0 MFMAs, the next tap's     \
           under the kk-1 MFMAs, +8 VGPRs = 2 instead of 3 wavefronts per SIMD on the (2, 1, 4)    \
           variant: 0.321-0.335 against 0.307-0.319 ms */                                          \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        const unsigned char *ab = s_a[0] + cur * (kPatch * kCvRow) + (dy * PWl + dx) * kCvRow;      \
        if constexpr (PIPE == 2) {                                                                  \
            constexpr int ndy = ((TAP) + 1) % 9 / 3, ndx = ((TAP) + 1) % 9 - ndy * 3;               \
            const unsigned char *abn = s_a[0] + ((TAP) == 8 ? (cur ^ 1) : cur) * (kPatch * kCvRow) + (ndy * PWl + ndx) * kCvRow; \
            const bf16x8 g00 = __builtin_bit_cast(bf16x8, X0), g10 = __builtin_bit_cast(bf16x8, X2); \
            const bf16x8 g01 = __builtin_bit_cast(bf16x8, X1), g11 = __builtin_bit_cast(bf16x8, X3); \
            /* pair p = (kk, block i) uses ring slot p & 3; in front of it the fragment of pair p + 3 \
               is requested (pairs 8, 9, 10 = the next step's first three; with two blocks a step  \
               has four pairs and pairs 4, 5, 6 are the next step's) */                             \
            if constexpr (MB == 2) {                                                                \
            CV_PAIR(fr3, ab + a_off[1 % MB] + 32, fr0, 0, g00, g10)                                 \
            CV_PAIR(fr0, abn + a_off[0], fr1, 1 % MB, g00, g10)                                     \
            CV_PAIR(fr1, abn + a_off[1 % MB], fr2, 0, g01, g11)                                     \
            CV_PAIR(fr2, abn + a_off[0] + 32, fr3, 1 % MB, g01, g11)                                \
            } else {                                                                                \
            CV_PAIR(fr3, ab + a_off[3 % MB], fr0, 0, g00, g10)                                      \
            CV_PAIR(fr0, ab + a_off[0] + 32, fr1, 1 % MB, g00, g10)                                 \
            CV_PAIR(fr1, ab + a_off[1 % MB] + 32, fr2, 2 % MB, g00, g10)                            \
            CV_PAIR(fr2, ab + a_off[2 % MB] + 32, fr3, 3 % MB, g00, g10)                            \
            CV_PAIR(fr3, ab + a_off[3 % MB] + 32, fr0, 0, g01, g11)                                 \
            CV_PAIR(fr0, abn + a_off[0], fr1, 1 % MB, g01, g11)                                     \
            CV_PAIR(fr1, abn + a_off[1 % MB], fr2, 2 % MB, g01, g11)                                \
            CV_PAIR(fr2, abn + a_off[2 % MB], fr3, 3 % MB, g01, g11)                                \
            }                                                                                       \
        } else {                                                                                    \
            CV_MFMA(0, X0, X2)                                                                      \
            CV_MFMA(1, X1, X3)                                                                      \
        }                                                                                           \
        if ((TAP) == 6) CV_STORE_A(cur ^ 1)                                                         \
        /* PIPE: (tap 7) the next patch is complete before tap 8 prefetches from it; (tap 2) nobody \
           reads the other buffer's old content any more before tap 6 overwrites it */              \
        if (PIPE && ((TAP) == 7 || (TAP) == 2)) __syncthreads();                                    \
    }
    int cur = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int step0 = chunk * 9;
        CV_STEP(0, bx0, bx1, bx2, bx3, bz0, bz1, bz2, bz3)
        CV_STEP(1, by0, by1, by2, by3, bx0, bx1, bx2, bx3)
        CV_STEP(2, bz0, bz1, bz2, bz3, by0, by1, by2, by3)
        CV_STEP(3, bx0, bx1, bx2, bx3, bz0, bz1, bz2, bz3)
        CV_STEP(4, by0, by1, by2, by3, bx0, bx1, bx2, bx3)
        CV_STEP(5, bz0, bz1, bz2, bz3, by0, by1, by2, by3)
        CV_STEP(6, bx0, bx1, bx2, bx3, bz0, bz1, bz2, bz3)
        CV_STEP(7, by0, by1, by2, by3, bx0, bx1, bx2, bx3)
        CV_STEP(8, bz0, bz1, bz2, bz3, by0, by1, by2, by3)
        // nine steps = three rotations of the sets: the names are back where they started
        if (!PIPE) __syncthreads();   // every wavefront is done with patch `cur`, the next one is stored
        cur ^= 1;
    }
#undef CV_STEP
#undef CV_PAIR
#undef CV_MFMA
#undef CV_LOAD_A
#undef CV_STORE_A
#undef CV_LOAD_B

    // ---- epilogue (ia_conv3.hpp)
    conv3_store_tile<MB>(acc, a, a.y[grp][lv] + (size_t)b * H * W * a.ys, wm * kWM, nt * kCvBN + wn * 64,
                         grp * a.Cout, H, W, TW, y0, x0, tile_px, lane);
}

// weights (groups * Cout, 3, 3, Cin) bf16 -> [groups][ceil(Cout / 256)][Cin / 32][9] steps of 256 x 32
// values in FRAGMENT ORDER [wn 4][j 2][kk 2][lane 64][8]: output channel wn * 64 + j * 32 + (lane & 31),
// k = kk * 16 + (lane >> 5) * 8 + e -- what lane `lane` of wavefront column wn feeds into the MFMA
// (j, kk) of the step, so that a wavefront's fragment load is one contiguous KiB.  Output channels
// beyond Cout are zero.
__global__ void __launch_bounds__(256) k_conv3x3_pack(const uint16_t *w, uint16_t *wp, int Cin, int Cout, int groups)
{
    const int ntile = (Cout + kCvBN - 1) / kCvBN, nchunk = Cin / kCvBK;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)groups * ntile * kCvBN * 9 * Cin;
    if (idx >= total) return;
    const int e = (int)(idx & 7);
    int64_t r = idx >> 3;
    const int lane = (int)(r & 63); r >>= 6;
    const int kk = (int)(r & 1); r >>= 1;
    const int j = (int)(r & 1); r >>= 1;
    const int wn = (int)(r & 3); r >>= 2;
    const int tap = (int)(r % 9); r /= 9;
    const int chunk = (int)(r % nchunk); r /= nchunk;
    const int nt = (int)(r % ntile);
    const int g = (int)(r / ntile);
    const int n = wn * 64 + j * 32 + (lane & 31), k = kk * 16 + (lane >> 5) * 8 + e;
    const int co = nt * kCvBN + n;
    wp[idx] = co < Cout ? w[(((size_t)(g * Cout + co) * 9) + tap) * Cin + chunk * kCvBK + k] : (uint16_t)0;
}

// tile shape for an H x W map: TH * TW <= max_px, (TH + 2) * (TW + 16) <= max_rows LDS rows,
// (TH + 2) * (TW + 2) <= the patch pixels of the tile size; least overhang
static void conv3_tile_shape_search(int H, int W, int max_px, int max_rows, int &TH, int &TW);

// memoised per (H, W, tile size): the search walks ~3 800 shapes per level and launch
static void conv3_tile_shape(int H, int W, int max_px, int max_rows, int &TH, int &TW)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, std::pair<int, int>> memo;
    const auto key = std::make_tuple(H, W, max_px, max_rows);
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = memo.find(key);
        if (it != memo.end()) { TH = it->second.first; TW = it->second.second; return; }
    }
    conv3_tile_shape_search(H, W, max_px, max_rows, TH, TW);
    std::lock_guard<std::mutex> lock(mu);
    if (memo.size() > 4096) memo.clear();
    memo[key] = std::make_pair(TH, TW);
}

static void conv3_tile_shape_search(int H, int W, int max_px, int max_rows, int &TH, int &TW)
{
    double best = -1.0;
    TH = 4; TW = 16;
    const int max_patch = max_px == 128 ? kCvMaxPatchPx128 : kCvMaxPatchPx64;     // patch pixels a workgroup's threads can stage
    for (int tw = 4; tw <= 64; ++tw) {
        for (int th = 2; th <= 64; ++th) {
            if (th * tw > max_px || (th + 2) * (tw + kCvPitchPad) > max_rows ||
                (th + 2) * (tw + 2) > max_patch)
                continue;
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
    for (int l = 0; l < d->num_levels; ++l) {
        if (d->H[l] < 1 || d->W[l] < 1) return IA_E_ARG;
        for (int g = 0; g < d->groups; ++g)
            if (!d->x[g][l] || !d->y[g][l] || ((uintptr_t)d->x[g][l] & 15u) || ((uintptr_t)d->y[g][l] & 3u))
                return IA_E_ARG;
    }
    // Variant (MB, WM, WN): four wavefronts, tile = 32 * MB * WM pixels x 64 * WN channels.
    //   Cout > 128: (2, 1, 4), 64-pixel tiles, three workgroups per CU -- measured best or tied on every
    //   pyramid level against the 8-wavefront workgroups of round 3 (batch 16, 256 -> 256: 100 x 168
    //   0.319 ms against 0.368 / 0.389 for 256 / 128 pixels; 50 x 84 0.095 against 0.139 / 0.123;
    //   13 x 21 0.026 against 0.057 / 0.035); LARGE maps -- at least kCvBigTiles 128-pixel tiles in
    //   the batch -- take (4, 1, 4) in a launch of their own: half the weight-fragment traffic per
    //   pixel (100 x 168 at batch 16: 0.305 against 0.319 ms; 50 x 84: 0.112 against 0.095, stays);
    //   Cout <= 128: (2, 2, 2), 128-pixel tiles x 128 channels -- the ResNet stage-2 / stage-1
    //   bottlenecks (64 output channels: the second column of wavefronts multiplies zeros; a
    //   256-pixel x 64-channel tile would need six patch pieces per thread).
    // IA_CONV3_VARIANT = 41 / 21 forces (4, 1, 4) / (2, 1, 4) for every map where WN = 4 applies
    // (tools/time_conv3x3_bf16.py).
    const int wnc = d->cout <= 64 ? 1 : (d->cout <= 128 ? 2 : 4);
    const char *force = getenv("IA_CONV3_VARIANT");
    const int forced = (force && force[0] && force[1] == '1') ? force[0] - '0' : 0;
    // Cout <= 64: (1, 4, 1) -- four wavefronts of 32 pixels x 64 channels, no padded MFMAs (ResNet
    // stage 1, batch 16, 200 x 336: 0.168 ms against 0.269 for (2, 2, 2) and 0.246 for (1, 2, 2));
    // 64 < Cout <= 128 stays on (2, 2, 2) (100 x 168: 0.123 against 0.128 for (1, 2, 2)).  "22" forces
    // (2, 2, 2) for both, "12" forces (1, 2, 2) for 64 < Cout <= 128.
    const bool f22 = force && force[0] == '2' && force[1] == '2', f12 = force && force[0] == '1' && force[1] == '2';
    const bool narrow64 = (wnc == 1 && !f22) || (wnc == 2 && f12);
    hipStream_t st = (hipStream_t)stream;
    const int wnk = (wnc == 1 && !narrow64) ? 2 : wnc;         // "22" forced: 64 output channels on (2, 2, 2) as before
    for (int pass = 0; pass < 2; ++pass) {            // pass 0: the large maps on (4, 1, 4); pass 1: the rest
        const int mb = (pass == 0 && wnc == 4) ? 4 : (narrow64 ? 1 : 2);
        const int px = 32 * mb * (4 / wnk);
        const int rows = px == 128 ? ia::kCvMaxRows128 : ia::kCvMaxRows64;
        ia::Conv3Args a;
        memset(&a, 0, sizeof(a));
        a.wp = static_cast<const uint16_t *>(wp); a.bias = bias;
        a.B = d->batch; a.Cin = d->cin; a.Cout = d->cout; a.xs = d->x_stride; a.ys = d->y_stride;
        a.relu = relu ? 1 : 0; a.ntile = (d->cout + ia::kCvBN - 1) / ia::kCvBN;
        int64_t tiles = 0;
        int n = 0;
        for (int l = 0; l < d->num_levels; ++l) {
            int th, tw;
            ia::conv3_tile_shape(d->H[l], d->W[l], 128, ia::kCvMaxRows128, th, tw);
            const int64_t big_tiles = (int64_t)d->batch * ((d->H[l] + th - 1) / th) * ((d->W[l] + tw - 1) / tw);
            const bool big = wnc == 4 && (forced == 4 || (forced != 2 && big_tiles >= ia::kCvBigTiles));
            if (big != (pass == 0)) continue;
            a.H[n] = d->H[l]; a.W[n] = d->W[l];
            for (int g = 0; g < d->groups; ++g) {
                a.x[g][n] = static_cast<const uint16_t *>(d->x[g][l]);
                a.y[g][n] = static_cast<uint16_t *>(d->y[g][l]);
            }
            ia::conv3_tile_shape(a.H[n], a.W[n], px, rows, a.TH[n], a.TW[n]);
            a.tiles_y[n] = (a.H[n] + a.TH[n] - 1) / a.TH[n]; a.tiles_x[n] = (a.W[n] + a.TW[n] - 1) / a.TW[n];
            a.tile_off[n] = (int32_t)tiles;
            tiles += (int64_t)d->batch * a.tiles_y[n] * a.tiles_x[n];
            if (tiles > 2147483647LL) return IA_E_ARG;
            ++n;
        }
        if (n == 0) continue;
        a.L = n;
        for (int l = n; l <= IA_MAX_LEVELS; ++l) a.tile_off[l] = (int32_t)tiles;
        const dim3 grid((unsigned)tiles, (unsigned)(d->groups * a.ntile));
        // both Cout > 128 variants with the fragment ring (PIPE = 2; the (2, 1, 4) variant at 168 registers:
        // still three wavefronts per SIMD; 50 x 84 at batch 16 0.090 -> 0.087 ms, the backbone's 256 -> 256 /
        // 512 -> 512 layers 0.091 -> 0.088 / 0.097 -> 0.094)
        if (wnc == 4 && mb == 4) hipLaunchKernelGGL((ia::k_conv3x3_bf16<4, 1, 4, 2>), grid, dim3(256), 0, st, a);
        else if (wnc == 4) hipLaunchKernelGGL((ia::k_conv3x3_bf16<2, 1, 4, 2>), grid, dim3(256), 0, st, a);
        else if (mb == 1 && wnk == 1) hipLaunchKernelGGL((ia::k_conv3x3_bf16<1, 4, 1>), grid, dim3(256), 0, st, a);
        else if (mb == 1) hipLaunchKernelGGL((ia::k_conv3x3_bf16<1, 2, 2>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((ia::k_conv3x3_bf16<2, 2, 2, 2>), grid, dim3(256), 0, st, a);    // ring: 128 -> 128 at 100 x 168 0.095 -> 0.086 ms
        const int rc = ia::hip_status(hipGetLastError());
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
