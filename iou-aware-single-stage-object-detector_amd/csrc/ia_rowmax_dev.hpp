// Device code of the channels-last row-max wavefront, shared by decode.hip (k_rowmax_nhwc, one
// wavefront per workgroup) and select.hip (k_rowmax_filter_nhwc: the same wavefronts, four to a
// workgroup, followed in the SAME launch by the top-k filter workgroups).
#pragma once
#include <type_traits>
#include "ia_internal.hpp"
#include "ia_math.hpp"

namespace ia {

// max over `lanes` (a power of two) neighbouring lanes, valid in every lane of the group
__device__ __forceinline__ float lanes_max(float v, int lanes)
{
    for (int off = 1; off < lanes; off <<= 1) {
        const float o = __shfl_xor(v, off);
        v = (v < o) ? o : v;
    }
    return v;
}

// PPL positions per lane: one 16-byte load per class plane.  The logits are read
// exactly once, so the loads are non-temporal (no L2 / Infinity-Cache allocation):
// measured +9 % on the P3 stream (5.37 -> 5.85 TB/s, tools/ubench).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Lane;
template <> struct Lane<float> {
    static constexpr int PPL = 4;
    static __device__ __forceinline__ void load(const float *p, float (&v)[4])
    {
        f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
    static __device__ __forceinline__ void load_cached(const float *p, float (&v)[4])
    {
        f32x4 q = *reinterpret_cast<const f32x4 *>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
};
template <> struct Lane<uint16_t> {
    static constexpr int PPL = 8;
    static __device__ __forceinline__ void load(const uint16_t *p, float (&v)[8])
    {
        u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        v[0] = from_bits(q.x << 16); v[1] = from_bits(q.x & 0xffff0000u);
        v[2] = from_bits(q.y << 16); v[3] = from_bits(q.y & 0xffff0000u);
        v[4] = from_bits(q.z << 16); v[5] = from_bits(q.z & 0xffff0000u);
        v[6] = from_bits(q.w << 16); v[7] = from_bits(q.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void load_cached(const uint16_t *p, float (&v)[8])
    {
        u32x4 q = *reinterpret_cast<const u32x4 *>(p);
        v[0] = from_bits(q.x << 16); v[1] = from_bits(q.x & 0xffff0000u);
        v[2] = from_bits(q.y << 16); v[3] = from_bits(q.y & 0xffff0000u);
        v[4] = from_bits(q.z << 16); v[5] = from_bits(q.z & 0xffff0000u);
        v[6] = from_bits(q.w << 16); v[7] = from_bits(q.w & 0xffff0000u);
    }
};

struct RowmaxNhwcArgs {
    LevelTable t;
    ia_level_ptrs p;
    float *rowmax;
    int32_t blk_off[IA_MAX_LEVELS + 1];   // prefix of ceil(B * N_l / 64) over the levels in launch order
    int32_t batch, anchors_per_img, big_first;
    SelPlan plan;                         // top-k plan: group sizes / offsets of the group maxima
    uint32_t *groupmax;                   // group maxima as ordered keys (bits | 0x80000000: never 0)
};

constexpr int kMaxVpr = 32;              // 16-byte vectors per row: C * sizeof(T) <= 512 bytes
#ifndef IA_ROWMAX_BATCH
#define IA_ROWMAX_BATCH 20
#endif
constexpr int kRowmaxBatch = IA_ROWMAX_BATCH;   // vector loads in flight per lane (tools/ubench/rowmax_bench.hip)

// write-through (sc1) stores: visible to the other XCDs without a release fence once the storing
// wavefront has drained its memory counter (cdna_hip_programming.md, Guideline 16 R1)
typedef float ia_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt_b128(float *p, ia_f32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_wt_b32(float *p, float v)
{
    __hip_atomic_store(reinterpret_cast<uint32_t *>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// One wavefront: 64 consecutive rows of ONE image (unit `rem` of level `l`; ceil(N_l / 64) units
// per image, the last one partial).
//   PUBLISH = false: plain stores (the next kernel reads them);
//   PUBLISH = true : the scores go out as write-through stores, the wavefront drains its memory
//                    counter, and only then stores the group maxima of its rows -- non-zero words
//                    (ordered keys) that are the "these 64 rows are in memory" flags the filter
//                    workgroups of the same launch wait for (Guideline 16 R1: payload, drain, flag).
template <typename T, int VPR_T, bool PUBLISH, typename AT = RowmaxNhwcArgs>   // VPR_T = 0: run-time vectors per row
__device__ __forceinline__ void rowmax_nhwc_wave(const AT &a, int l, int rem, float *s_m, int lane)
{
    constexpr int PPL = Lane<T>::PPL;
    const int vpr = VPR_T ? VPR_T : a.t.C / PPL;
    const int n_l = a.t.anchor_off[l + 1] - a.t.anchor_off[l];
    const int upi = (n_l + 63) >> 6;                       // units per image
    const int b = rem / upi, u0 = (rem - b * upi) * 64;    // image, first anchor index of the unit
    const int64_t r0 = (int64_t)b * n_l + u0;              // row of the level's flat (B * N_l) space
    const int nrow = (n_l - u0 < 64) ? (n_l - u0) : 64;
    const int nvec = nrow * vpr;
    const T *src = static_cast<const T *>(a.p.cls[l]) + r0 * a.t.C;
    // this lane's row: its IoU logit is requested first so that its latency hides behind the
    // class loads instead of following the barrier
    const int64_t g = r0 + ((lane < nrow) ? lane : (nrow - 1));
    const float il = load_f32<T>(static_cast<const T *>(a.p.iou[l]) + g);
    // All loads of a batch are issued before the first one is consumed: written as one loop
    // (load, reduce, LDS store per vector) the compiler waits for each load before issuing the
    // next -- ONE kilobyte in flight per wavefront, a latency-bound kernel that only its 29
    // wavefronts per CU kept near 6 TB/s.
    auto batch = [&](int k0, auto nb_tag) {
        constexpr int NB = decltype(nb_tag)::value;
        float v[NB][PPL];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int f = (k0 + u) * 64 + lane;
            const int fc = (f < nvec) ? f : (nvec - 1);         // loads are never predicated
            Lane<T>::load(src + (size_t)fc * PPL, v[u]);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int f = (k0 + u) * 64 + lane;
            float m = v[u][0];
#pragma unroll
            for (int j = 1; j < PPL; ++j) m = (m < v[u][j]) ? v[u][j] : m;
            const int row = f / vpr, c4 = f - row * vpr;
            if (f < nvec) s_m[row * (vpr + 1) + c4] = m;
        }
    };
    if (VPR_T) {
        constexpr int NB = (VPR_T % kRowmaxBatch == 0) ? kRowmaxBatch : (VPR_T ? VPR_T : 1);
#pragma unroll
        for (int k = 0; k < VPR_T; k += NB) batch(k, std::integral_constant<int, NB>());
    } else {
        int k = 0;
        for (; k + 4 <= vpr; k += 4) batch(k, std::integral_constant<int, 4>());
        for (; k < vpr; ++k) batch(k, std::integral_constant<int, 1>());
    }
    // the tile belongs to this wavefront alone: its LDS writes only have to be complete
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float score = 0.0f;                                     // scores are >= 0
    const int i = (int)(g - (int64_t)b * n_l);
    float *dst = a.rowmax + (size_t)b * a.anchors_per_img + a.t.anchor_off[l] + i;
    if (lane < nrow) {
        const float *sr = s_m + lane * (vpr + 1);
        float m = sr[0];
        for (int c4 = 1; c4 < vpr; ++c4) m = (m < sr[c4]) ? sr[c4] : m;
#ifdef IA_ROWMAX_NOMATH                                      /* tools/ubench/rowmax_bench.hip only */
        score = m * il;
#else
        score = sqrt_sigmoidf_(m) * sqrt_sigmoidf_(il);
#endif
    }
    const int grp = a.plan.grp[l];
    const bool pub = PUBLISH && grp != 0;                   // only filtered levels have readers in this launch
    if (!pub) {
        if (lane < nrow) *dst = score;
    } else {
        // four lanes' scores in one 16-byte write-through store where the four rows are live,
        // contiguous (same image) and 16-byte aligned; single words otherwise
        const int q = lane & ~3;
        ia_f32x4 v4;
        v4.x = __shfl(score, q); v4.y = __shfl(score, q + 1); v4.z = __shfl(score, q + 2); v4.w = __shfl(score, q + 3);
        const bool quad = q + 3 < nrow;                      // (a unit lies inside one image)
        const bool aligned = quad && ((reinterpret_cast<uintptr_t>(a.rowmax + (size_t)b * a.anchors_per_img + a.t.anchor_off[l] +
                                                                    (size_t)(u0 + q)) & 15u) == 0);
#ifdef IA_ABL_PLAIN_STORE                                    /* tools/ubench/stage_bench.hip only */
        if (lane < nrow) *dst = score;
#else
        if (aligned) {
            if (lane == q) store_wt_b128(dst, v4);
        } else if (lane < nrow) {
            store_wt_b32(dst, score);
        }
#endif
    }
    if (a.groupmax && grp) {
        // maxima of groups of grp consecutive anchors of this image, as ordered keys (score bits |
        // sign bit: scores are >= +0, so the word is never 0); u0 is a multiple of 64, so the
        // groups are lane-aligned; ceil(N_l / grp) words per image
        const uint32_t key = __builtin_bit_cast(uint32_t, lanes_max(score, grp)) | 0x80000000u;
        if (pub) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // scores are in memory
        if ((lane & (grp - 1)) == 0 && lane < nrow) {
            uint32_t *gp = a.groupmax + a.plan.goff[l] + (size_t)b * ((n_l + grp - 1) / grp) + (u0 + lane) / grp;
            if (pub) __hip_atomic_store(gp, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *gp = key;
        }
    }
}

}  // namespace ia
