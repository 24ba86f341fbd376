// 3x3 / stride 1 / pad 1 convolution on bf16 channels-last tensors, the LARGE-map variant: a
// 256-pixel x 256-channel workgroup tile, eight wavefronts in two groups that alternate between an
// LDS-read slot and an MFMA slot ("ping-pong"), both operands staged global -> LDS by the DMA path
// (global_load_lds_dwordx4), no operand ever passes through staging registers.
// Same contract, packed-weight format and epilogue as k_conv3x3_bf16 (conv3x3_bf16.hip; reference
// iou_aware_retina_head.py:171-219, conv_module.py:149-163, resnet.py:215-255).
//
// Why (round-4 profile of k_conv3x3_bf16<4,1,4>, 128 pixels x 256 channels, two workgroups per CU):
// the matrix pipe was 0.55 busy.  Every K step opened with `ds_read_b128 x2; s_waitcnt; v_mfma`:
// the LDS latency of the A fragments sat in front of the MFMAs of every step and only the second
// wavefront of the SIMD -- stalled on the same pattern -- could cover it; every workgroup fetched
// all 1.2 MB of weights from L2 for 128 pixels (32 B/clk/CU at full MFMA rate, half of the L2's
// bandwidth), and three register sets of weight fragments plus the patch staging registers held
// the kernel at 256 VGPRs.
//
// This kernel:
//  * tile = TH x TW <= 256 pixels x 256 output channels per workgroup: the weights are fetched once
//    per 256 pixels (16 B/clk/CU at full rate);
//  * wavefront w = (group g = w >> 2, column wn = w & 3): 128 pixels x 64 channels, 4 x 2
//    accumulator blocks of 32 x 32 (v_mfma_f32_32x32x16_bf16), 128 accumulator registers.
//    Wavefronts w and w + 4 share a SIMD (a workgroup's wavefronts go to the SIMDs cyclically), so
//    every SIMD holds one wavefront of each group;
//  * K loop = Cin / 32 chunks x 9 taps.  A step (chunk, tap) of a wavefront is two SLOTS separated by
//    workgroup barriers: a LOAD slot (address arithmetic, 8 A + 4 B `ds_read_b128`, this wavefront's
//    share of the DMA requests for the weights three steps ahead and the next chunk's halo patch,
//    `s_waitcnt` for the reads and for the DMA pieces requested two steps ago) and an MFMA slot (16
//    MFMAs at raised priority).  Group 1 runs one slot behind group 0: while one wavefront of a SIMD
//    is in its MFMA slot (16 x 32 = 512 cycles of the pipe) the other one reads its fragments, so the
//    pipe always finds a wavefront whose operands are in registers;
//  * halo patch of a 32-channel chunk: (TH + 2) x (TW + 2) pixels x 64 B in LDS, double-buffered,
//    row pitch TW + 16 pixels (LDS row R of tile pixel m is then = m + const (mod 16) for every tap)
//    and the 16-byte pieces of a row XOR-swizzled by (R >> 2) & 3: the 16 lanes of every
//    ds_read_b128 service group fall into 16 different bank slots.  The DMA writes LDS linearly
//    (wave base + lane x 16), so the swizzle is applied on the SOURCE side: lane i of a piece
//    fetches the 16 bytes that belong into its slot.  Padding (pixels outside the image) and the
//    pitch gap are zeroed once per workgroup and never written by the DMA (those lanes are masked
//    off): the out-of-image positions of a tile are the same for every chunk;
//  * weights: the fragment-order packing of ia_conv3x3_bf16_pack, a step's 256 x 32 block = 16 KiB
//    contiguous = 16 DMA pieces of 1 KiB, ring of four steps in LDS; both groups read the same block.
// LDS: 2 x 36 KiB (patches) + 64 KiB (weights) = 136 KiB, one workgroup per CU.
#include <stdlib.h>
#include <string.h>
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_conv3.hpp"

namespace ia {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kPpPatchBytes = kPpMaxRows * 64;       // one patch buffer
constexpr int kPpBSlot = 256 * 32 * 2;               // one step's weights
constexpr int kPpBRing = 4;
constexpr int kPpPieces = (kPpMaxRows / 16 + 7) / 8; // patch DMA pieces per wavefront (16 LDS rows each)
static_assert(kPpPieces == 5, "the tap schedule below places five pieces");

__device__ __forceinline__ uint32_t pp_bf16_rne(float f)
{
    uint32_t u = to_bits(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// one DMA piece: 64 lanes x 16 bytes, LDS destination = dst + lane * 16 (dst wave-uniform)
__device__ __forceinline__ void dma16(const void *src, unsigned char *dst)
{
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
}

typedef int v4i __attribute__((ext_vector_type(4)));
// LDS reads as inline assembly: behind a pending LDS-DMA request the compiler waits vmcnt(0) in front
// of every ds_read it knows about (it cannot tell the DMA's destination from the read's source), which
// would drain the prefetch pipeline in every slot.  The reads' completion is covered by
// PP_WAIT_FRAGS, which names every fragment register as an operand so that no MFMA can be scheduled
// in front of it.
#define PP_LDS_READ(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define PP_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define PP_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

__global__ void __launch_bounds__(512, 2) k_conv3x3_bf16_pp(Conv3Args a)
{
    __shared__ __attribute__((aligned(1024))) unsigned char s_patch[2][kPpPatchBytes];
    __shared__ __attribute__((aligned(1024))) unsigned char s_b[kPpBRing][kPpBSlot];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int grp_w = wv >> 2, wn = wv & 3;                 // wavefront-uniform
    int lv = 0;
    while (lv + 1 < a.L && (int)blockIdx.x >= a.tile_off[lv + 1]) ++lv;
    int t = (int)blockIdx.x - a.tile_off[lv];
    const int H = a.H[lv], W = a.W[lv], TH = a.TH[lv], TW = a.TW[lv];
    const int txi = t % a.tiles_x[lv]; t /= a.tiles_x[lv];
    const int tyi = t % a.tiles_y[lv];
    const int b = t / a.tiles_y[lv];
    const int grp = (int)blockIdx.y / a.ntile, nt = (int)blockIdx.y - grp * a.ntile;
    const int y0 = tyi * TH, x0 = txi * TW;
    const int P = TW + 16, tile_px = TH * TW;
    const int nchunk = a.Cin / 32, nsteps = nchunk * 9;

    // ---- zero both patch buffers (padding / pitch gap stay zero: the DMA never writes them)
    {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (int o = tid * 16; o < 2 * kPpPatchBytes; o += 512 * 16)
            *reinterpret_cast<uint4 *>(&s_patch[0][0] + o) = z;
    }

    // ---- this lane's A rows: LDS row of pixel m at tap (0, 0)
    int r0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        int m = grp_w * 128 + mb * 32 + (lane & 31);
        m = m < tile_px ? m : tile_px - 1;                  // idle rows read a valid address
        const int ty = m / TW, tx = m - ty * TW;
        r0[mb] = ty * P + tx;
    }
    const int khalf = lane >> 5;                            // which 8 of a k-block's 16 values

    // ---- patch DMA pieces of this wavefront: piece q = u * 8 + wv covers LDS rows 16 q .. 16 q + 15,
    // lane -> row 16 q + (lane >> 2), LDS slot lane & 3 <- source 16-byte part (lane & 3) ^ ((R >> 2) & 3)
    const uint16_t *xb = a.x[grp][lv] + (size_t)b * H * W * a.xs;
    const int nrows = (TH + 2) * P;
    int po[kPpPieces];                                      // element offset inside the image, < 0: no request
#pragma unroll
    for (int u = 0; u < kPpPieces; ++u) {
        const int R = (u * 8 + wv) * 16 + (lane >> 2);
        const int py = R / P, px = R - py * P;
        const int iy = y0 + py - 1, ix = x0 + px - 1;
        const bool ok = R < nrows && px < TW + 2 && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const int part = (lane & 3) ^ ((R >> 2) & 3);
        po[u] = ok ? (iy * W + ix) * a.xs + part * 8 : -1;
    }
    // (a piece whose 64 lanes are all masked is not issued at all -- the compiler branches around it --
    // so the wait counts below only ever count the WEIGHT pieces, which every slot issues: the
    // patch pieces in flight make the wait stricter, never weaker)

    // packed weights: [group][ntile][step][16 KiB]; this wavefront's two pieces of a step
    const uint16_t *wsrc = a.wp + (size_t)(grp * a.ntile + nt) * nsteps * (256 * 32) + (size_t)wv * 512 + lane * 8;

#define PP_DMA_B(step)                                                                             \
    {                                                                                              \
        const int sB = (step) < nsteps ? (step) : nsteps - 1;                                      \
        const uint16_t *src = wsrc + (size_t)sB * (256 * 32);                                      \
        unsigned char *dst = &s_b[(step) & (kPpBRing - 1)][0] + wv * 1024;                         \
        dma16(src, dst);                                                                           \
        dma16(src + 8 * 512, dst + 8 * 1024);                                                      \
    }
#define PP_DMA_A(u, chunk_, buf)                                                                   \
    {                                                                                              \
        const int cA = (chunk_) < nchunk ? (chunk_) : nchunk - 1;                                  \
        if (po[u] >= 0)                                                                            \
            dma16(xb + (size_t)po[u] + cA * 32, &s_patch[buf][0] + ((u) * 8 + wv) * 1024);         \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- prologue: patch of chunk 0, weights of steps 0, 1, 2
    __syncthreads();                                        // zero fill complete before the DMA lands
    PP_DMA_A(0, 0, 0) PP_DMA_A(1, 0, 0) PP_DMA_A(2, 0, 0) PP_DMA_A(3, 0, 0) PP_DMA_A(4, 0, 0)
    PP_DMA_B(0) PP_DMA_B(1) PP_DMA_B(2)
    PP_WAIT_VM(0);
    PP_BARRIER();
    if (grp_w == 1) PP_BARRIER();                           // group 1 runs one slot behind

    v4i fa[4][2], fb[2][2];
    // One step.  LOAD slot: fragments of (chunk, TAP) from LDS; DMA requests: weights of step + 3 (two
    // pieces), and at taps 0..4 one piece of the next chunk's patch.  The wait leaves at most NVM = 4
    // requests in flight = the weight pieces of this slot and of the previous one (in-order return:
    // with patch pieces among the youngest it waits for more, never less): everything older -- the
    // weights of step + 1 and, from tap 6 on, the whole next patch -- has landed when the barrier is
    // passed, one full step before it is read.
#define PP_STEP(TAP, NVM)                                                                          \
    {                                                                                              \
        constexpr int dy = (TAP) / 3, dx = (TAP) - dy * 3;                                         \
        const int step = chunk * 9 + (TAP);                                                        \
        const uint32_t pa = (uint32_t)(uintptr_t)(lptr_t)&s_patch[cur][0];                         \
        const uint32_t pb = (uint32_t)(uintptr_t)(lptr_t)&s_b[step & (kPpBRing - 1)][0] + wn * 4096 + lane * 16; \
        const int toff = dy * P + dx;                                                              \
        /* the fragment addresses of all 9 taps are loop-invariant: left alone the compiler hoists  \
           36 of them out of the chunk loop and spills (scratch loads would also count in vmcnt) */ \
        asm volatile("" : "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(r0[3]));                     \
        _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) {                                         \
            const int R = r0[mb] + toff;                                                           \
            const int o0 = (R << 6) + ((((R >> 2) ^ khalf) & 3) << 4);                             \
            PP_LDS_READ(fa[mb][0], pa + o0, 0);                                                    \
            PP_LDS_READ(fa[mb][1], pa + (o0 ^ 32), 0);                                             \
        }                                                                                          \
        PP_LDS_READ(fb[0][0], pb, 0);                                                              \
        PP_LDS_READ(fb[0][1], pb, 1024);                                                           \
        PP_LDS_READ(fb[1][0], pb, 2048);                                                           \
        PP_LDS_READ(fb[1][1], pb, 3072);                                                           \
        PP_DMA_B(step + 3)                                                                         \
        if ((TAP) < kPpPieces) PP_DMA_A((TAP) < kPpPieces ? (TAP) : 0, chunk + 1, cur ^ 1)         \
        PP_WAIT_VM(NVM);                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                        \
                     : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[2][0]), "+v"(fa[2][1]), \
                       "+v"(fa[3][0]), "+v"(fa[3][1]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1])  \
                     :: "memory");                                                                 \
        PP_BARRIER();                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                             \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                           \
            _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) {                                     \
                acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[mb][kk]), __builtin_bit_cast(bf16x8, fb[0][kk]), acc[mb][0], 0, 0, 0); \
                acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[mb][kk]), __builtin_bit_cast(bf16x8, fb[1][kk]), acc[mb][1], 0, 0, 0); \
            }                                                                                      \
        __builtin_amdgcn_s_setprio(0);                                                             \
        PP_BARRIER();                                                                              \
    }
    int cur = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        PP_STEP(0, 4)
        PP_STEP(1, 4)
        PP_STEP(2, 4)
        PP_STEP(3, 4)
        PP_STEP(4, 4)
        PP_STEP(5, 4)
        PP_STEP(6, 4)
        PP_STEP(7, 4)
        PP_STEP(8, 4)
        cur ^= 1;
    }
#undef PP_STEP
#undef PP_DMA_A
#undef PP_DMA_B
    if (grp_w == 0) PP_BARRIER();                           // both groups: the same number of barriers
    PP_WAIT_VM(0);                                          // the clamped requests of the last steps

    // ---- epilogue: C block (i, j): column n = lane & 31, row m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int odd = lane & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nt * 256 + wn * 64 + j * 32 + (lane & 31);          // channel inside the group
        const bool n_ok = (n - odd) + 1 < a.Cout;                         // the pair this lane stores (Cout is even)
        const float bz = (a.bias && n < a.Cout) ? a.bias[grp * a.Cout + n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float v0 = acc[i][j][r] + bz, v1 = acc[i][j][r + 1] + bz;
                if (a.relu) { v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; }
                // even lane keeps row r and takes the odd neighbour's row-r value (channel n + 1);
                // odd lane keeps row r + 1 and takes the even neighbour's (channel n - 1)
                const float give = odd ? v0 : v1;
                const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
                const int rr = odd ? r + 1 : r;
                const int m = grp_w * 128 + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
                const uint32_t lo = pp_bf16_rne(odd ? got : v0), hi = pp_bf16_rne(odd ? v1 : got);
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

int launch_conv3x3_bf16_pp(const Conv3Args &a, dim3 grid, hipStream_t st)
{
    hipLaunchKernelGGL(k_conv3x3_bf16_pp, grid, dim3(512), 0, st, a);
    return hip_status(hipGetLastError());
}

}  // namespace ia
