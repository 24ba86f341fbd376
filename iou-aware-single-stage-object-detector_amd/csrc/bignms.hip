// mmdet.ops.nms.nms for MORE than IA_MAX_CANDIDATES boxes (reference mmdet/ops/nms/src/nms_cpu.cpp:4-59
// takes any n; nms.hip's single-problem path keeps the n x n suppression bit matrix and an LDS sort,
// hence its 8192-box limit).  Same semantics, chunked:
//   1. keys ordered(score) << 32 | ~index, sorted descending device-wide (rocPRIM radix sort: a plain
//      library sort, like the library GEMMs) -> the canonical order (score desc, index asc);
//   2. the sorted boxes are taken kS = 8192 at a time: a box of the chunk is dropped if one of the
//      boxes KEPT in earlier chunks suppresses it (k_big_prefilter: 64 x 64 tiles over chunk x kept
//      list, the exact test of ia_nms.hpp); the survivors go through nms.hip's single-problem NMS
//      (dropped boxes are moved far away with score -inf: they can neither suppress nor outrank a
//      live box) and what it keeps is appended to the kept list;
//   3. kept boxes are emitted in ascending input index (nms_cpu.cpp:58).
// Work: every pair (box, earlier kept box) is tested once, spread over the chip; no host
// synchronisation (the kept count stays on the device; the prefilter grid is sized for the worst
// case and its surplus workgroups exit at once).
#include <rocprim/device/device_radix_sort.hpp>
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_nms.hpp"

namespace ia {

constexpr int kS = IA_MAX_CANDIDATES;          // boxes per chunk (the single-problem path's size)

struct BigLayout { size_t keys_in, keys_out, sort_tmp, sbox, sscore, supp, kept, kept_count, keepflag, cdets, ckeep, ccount, inner, total; };

// Reserved for the sort's temporary storage: a second key buffer plus its histograms / look-back
// state, with slack.  A fixed formula (not rocPRIM's size query, which needs a device) so that
// ia_nms_workspace_bytes is a pure function of n; launch_nms_big checks the real need against it.
static size_t sort_temp_bytes(int n) { return (size_t)n * 16 + ((size_t)4 << 20); }

static BigLayout big_layout(int n)
{
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    BigLayout w;
    size_t o = 0;
    const size_t N = (size_t)n;
    w.keys_in = o; o = up(o + N * 8);
    w.keys_out = o; o = up(o + N * 8);
    w.sort_tmp = o; o = up(o + sort_temp_bytes(n));
    w.sbox = o; o = up(o + N * 16);
    w.sscore = o; o = up(o + N * 4);
    w.supp = o; o = up(o + N * 4);
    w.kept = o; o = up(o + N * 4);
    w.kept_count = o; o = up(o + 256);
    w.keepflag = o; o = up(o + N);
    w.cdets = o; o = up(o + (size_t)kS * 5 * 4);
    w.ckeep = o; o = up(o + (size_t)kS * 4);
    w.ccount = o; o = up(o + 256);
    w.inner = o; o = up(o + nms_single_workspace_bytes(kS));
    w.total = o;
    return w;
}

size_t nms_big_workspace_bytes(int n) { return n < 1 ? 0 : big_layout(n).total; }

__global__ void __launch_bounds__(256) k_big_keys(const float *dets, int n, uint64_t *keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = ((uint64_t)ordered_key(dets[5 * (size_t)i + 4]) << 32) | (uint64_t)(0xffffffffu - (uint32_t)i);
}

__global__ void __launch_bounds__(256) k_big_gather(const float *dets, const uint64_t *keys, int n, float4 *sbox,
                                                    float *sscore, uint32_t *supp, uint32_t *kept_count,
                                                    uint8_t *keepflag)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j == 0) *kept_count = 0;
    if (j >= n) return;
    const uint32_t idx = 0xffffffffu - (uint32_t)keys[j];
    const float *d = dets + 5 * (size_t)idx;
    sbox[j] = make_float4(d[0], d[1], d[2], d[3]);
    sscore[j] = d[4];
    supp[j] = 0;
    keepflag[j] = 0;
}

// chunk boxes [base, base + m) against the kept list: blockIdx.y = candidate tile, blockIdx.x = kept tile
__global__ void __launch_bounds__(64) k_big_prefilter(const float4 *sbox, const uint32_t *kept, const uint32_t *kept_count,
                                                      int base, int m, IouThr thr, uint32_t *supp)
{
    const uint32_t nk = *kept_count;
    const uint32_t k0 = blockIdx.x * 64u;
    if (k0 >= nk) return;                                     // surplus workgroup
    const int lane = threadIdx.x;
    const int ci = blockIdx.y * 64 + lane;
    const bool cok = ci < m;
    const float4 c = sbox[base + (cok ? ci : m - 1)];
    const float car = ((c.z - c.x) + 1.0f) * ((c.w - c.y) + 1.0f);
    const bool kok = k0 + lane < nk;
    const float4 s = sbox[kept[kok ? k0 + lane : nk - 1]];
    const float sar = ((s.z - s.x) + 1.0f) * ((s.w - s.y) + 1.0f);
    uint64_t todo = __ballot(kok);
    bool hit = false;
    while (todo) {
        const int kk = __builtin_ctzll(todo);
        todo &= todo - 1;
        auto bc = [kk](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), kk)); };
        if (suppresses(bc(s.x), bc(s.y), bc(s.z), bc(s.w), bc(sar), c.x, c.y, c.z, c.w, car, thr)) hit = true;
    }
    if (cok && hit) supp[base + ci] = 1u;
}

// (m, 5) dets of the chunk for the single-problem NMS; boxes dropped by the prefilter become inert
__global__ void __launch_bounds__(256) k_big_chunk_dets(const float4 *sbox, const float *sscore, const uint32_t *supp,
                                                        int base, int m, float *cdets)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    float4 b = sbox[base + i];
    float sc = sscore[base + i];
    if (supp[base + i]) { b = make_float4(-1e8f, -1e8f, -1e8f, -1e8f); sc = -__builtin_inff(); }
    float *d = cdets + 5 * (size_t)i;
    d[0] = b.x; d[1] = b.y; d[2] = b.z; d[3] = b.w; d[4] = sc;
}

// what the chunk's NMS kept (ascending chunk positions), minus the inert boxes -> kept list, keep flags
__global__ void __launch_bounds__(256) k_big_append(const int32_t *ckeep, const int32_t *ccount, const uint32_t *supp,
                                                    const uint64_t *keys, int base, uint32_t *kept,
                                                    uint32_t *kept_count, uint8_t *keepflag)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    const int cnt = *ccount;
    if (threadIdx.x == 0) s_base = *kept_count;
    __syncthreads();
    for (int i0 = 0; i0 < cnt; i0 += 256) {
        const int i = i0 + (int)threadIdx.x;
        const int p = i < cnt ? ckeep[i] : 0;
        const bool live = i < cnt && !supp[base + p];
        const uint64_t mk = __ballot(live);
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) s_wave[wv] = (uint32_t)__builtin_popcountll(mk);
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (int w = 0; w < 4; ++w) { before += (w < wv) ? s_wave[w] : 0u; all += s_wave[w]; }
        if (live) {
            const uint32_t lo = (uint32_t)mk, hi = (uint32_t)(mk >> 32);
            const uint32_t pre = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
            kept[s_base + before + pre] = (uint32_t)(base + p);
            keepflag[0xffffffffu - (uint32_t)keys[base + p]] = 1;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) *kept_count = s_base;
}

// kept input indices, ascending (nms_cpu.cpp:58)
__global__ void __launch_bounds__(1024) k_big_emit(const uint8_t *keepflag, int n, int32_t *keep, int32_t *count)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + (int)threadIdx.x;
        const bool kp = i < n && keepflag[i];
        const uint64_t mk = __ballot(kp);
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) s_wave[wv] = (uint32_t)__builtin_popcountll(mk);
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (int w = 0; w < 16; ++w) { before += (w < wv) ? s_wave[w] : 0u; all += s_wave[w]; }
        if (kp) {
            const uint32_t lo = (uint32_t)mk, hi = (uint32_t)(mk >> 32);
            keep[s_base + before + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u))] = i;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = (int32_t)s_base;
}

int launch_nms_big(const float *dets, int n, float iou_thr, int32_t *keep, int32_t *count, void *workspace,
                   size_t workspace_bytes, hipStream_t s)
{
    if (n < 1 || !dets || !keep || !count || !workspace) return IA_E_ARG;
    const BigLayout w = big_layout(n);
    if (workspace_bytes < w.total) return IA_E_WORKSPACE;
    char *ws = static_cast<char *>(workspace);
    uint64_t *keys_in = reinterpret_cast<uint64_t *>(ws + w.keys_in), *keys = reinterpret_cast<uint64_t *>(ws + w.keys_out);
    float4 *sbox = reinterpret_cast<float4 *>(ws + w.sbox);
    float *sscore = reinterpret_cast<float *>(ws + w.sscore);
    uint32_t *supp = reinterpret_cast<uint32_t *>(ws + w.supp), *kept = reinterpret_cast<uint32_t *>(ws + w.kept);
    uint32_t *kept_count = reinterpret_cast<uint32_t *>(ws + w.kept_count);
    uint8_t *keepflag = reinterpret_cast<uint8_t *>(ws + w.keepflag);
    float *cdets = reinterpret_cast<float *>(ws + w.cdets);
    int32_t *ckeep = reinterpret_cast<int32_t *>(ws + w.ckeep), *ccount = reinterpret_cast<int32_t *>(ws + w.ccount);
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_big_keys, dim3(nb), dim3(256), 0, s, dets, n, keys_in);
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_keys_desc(nullptr, need, keys_in, keys, (size_t)n, 0, 64, s);
    if (e != hipSuccess) return hip_status(e);
    if (need > w.sbox - w.sort_tmp) return IA_E_WORKSPACE;
    e = rocprim::radix_sort_keys_desc(ws + w.sort_tmp, need, keys_in, keys, (size_t)n, 0, 64, s);
    if (e != hipSuccess) return hip_status(e);
    hipLaunchKernelGGL(k_big_gather, dim3(nb), dim3(256), 0, s, dets, keys, n, sbox, sscore, supp, kept_count, keepflag);
    const IouThr thr = make_thr(iou_thr);
    for (int base = 0; base < n; base += kS) {
        const int m = (n - base < kS) ? (n - base) : kS;
        if (base > 0)
            hipLaunchKernelGGL(k_big_prefilter, dim3((unsigned)((base + 63) / 64), (unsigned)((m + 63) / 64)), dim3(64), 0, s,
                               sbox, kept, kept_count, base, m, thr, supp);
        hipLaunchKernelGGL(k_big_chunk_dets, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, sbox, sscore, supp, base, m,
                           cdets);
        int rc = launch_nms_single(cdets, m, iou_thr, ckeep, ccount, ws + w.inner, w.total - w.inner, s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_big_append, dim3(1), dim3(256), 0, s, ckeep, ccount, supp, keys, base, kept, kept_count, keepflag);
    }
    hipLaunchKernelGGL(k_big_emit, dim3(1), dim3(1024), 0, s, keepflag, n, keep, count);
    return hip_status(hipGetLastError());
}

}  // namespace ia
