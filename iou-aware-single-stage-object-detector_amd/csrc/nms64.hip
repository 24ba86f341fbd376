// mmdet.ops.nms.nms on DOUBLE boxes: nms_cpu_kernel<double> (reference mmdet/ops/nms/src/nms_cpu.cpp:4-59,
// instantiated for float AND double by AT_DISPATCH_FLOATING_TYPES, :63).  The fp32 entry (ia_nms) cannot
// stand in for it: areas, intersections and the quotient are evaluated in the tensor's own type, and
// `ovr >= threshold` compares a double quotient with the float threshold PROMOTED to double -- for the
// boxes of the `>=` corner case (IoU exactly 1/3) and threshold float(1/3) = 0.33333334 the fp32
// instantiation suppresses, the fp64 one does not.
//
// A rarely used type, so the plain formulation, on the device, without host round trips:
//   k_nms64_keys     order-preserving 64-bit keys of the scores
//   rocPRIM          stable descending radix sort of (key, index) pairs: equal scores keep ascending
//                    input order (the reference's std::sort-based order of ties is unspecified)
//   k_nms64_adj      upper triangle of the n x n relation "IoU(i, j) >= thr" over the SORTED boxes,
//                    64 x 64 bit tiles, fp64 arithmetic in the reference's operation order
//   k_nms64_resolve  one workgroup walks the sorted list: a box not yet removed is kept and ORs its
//                    row into the removed set (LDS bitmap)
//   k_nms64_emit     kept input indices in ascending order (nms_cpu.cpp:58) + their count
#include <rocprim/device/device_radix_sort.hpp>
#include "ia_internal.hpp"

namespace ia {

constexpr int kNms64Max = 16384;          // n x n / 8 bytes of bit matrix: 32 MiB at the limit

__device__ __forceinline__ uint64_t ordered64(double d)
{
    const uint64_t u = (uint64_t)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__global__ void k_nms64_keys(const double *dets, int n, uint64_t *keys, uint32_t *vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = ordered64(dets[(size_t)i * 5 + 4]); vals[i] = (uint32_t)i; }
}

// tile (ti, tj), tj >= ti: thread r owns sorted row ti * 64 + r and tests the 64 columns of tile tj
__global__ void __launch_bounds__(64) k_nms64_adj(const double *dets, const uint32_t *order, int n, double thr,
                                                  int words, uint64_t *adj)
{
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj < ti) return;
    __shared__ double cb[64][5];
    const int r = threadIdx.x;
    const int col = tj * 64 + r;
    if (col < n) {
        const double *d = dets + (size_t)order[col] * 5;
        cb[r][0] = d[0]; cb[r][1] = d[1]; cb[r][2] = d[2]; cb[r][3] = d[3];
        cb[r][4] = (d[2] - d[0] + 1.0) * (d[3] - d[1] + 1.0);            // areas, nms_cpu.cpp:18
    }
    __syncthreads();
    const int row = ti * 64 + r;
    if (row >= n) return;
    const double *d = dets + (size_t)order[row] * 5;
    const double ix1 = d[0], iy1 = d[1], ix2 = d[2], iy2 = d[3];
    const double iarea = (ix2 - ix1 + 1.0) * (iy2 - iy1 + 1.0);
    uint64_t bits = 0;
    const int cols = (n - tj * 64) < 64 ? (n - tj * 64) : 64;
    for (int c = 0; c < cols; ++c) {
        if (tj * 64 + c <= row) continue;                                  // later boxes only (:41)
        const double xx1 = ix1 > cb[c][0] ? ix1 : cb[c][0];               // std::max, :46-49
        const double yy1 = iy1 > cb[c][1] ? iy1 : cb[c][1];
        const double xx2 = ix2 < cb[c][2] ? ix2 : cb[c][2];
        const double yy2 = iy2 < cb[c][3] ? iy2 : cb[c][3];
        double w = xx2 - xx1 + 1.0;  w = 0.0 > w ? 0.0 : w;                // :51-52
        double h = yy2 - yy1 + 1.0;  h = 0.0 > h ? 0.0 : h;
        const double inter = w * h;
        const double ovr = inter / (iarea + cb[c][4] - inter);             // :54
        if (ovr >= thr) bits |= 1ull << c;                                  // :55
    }
    adj[(size_t)row * words + tj] = bits;
}

__global__ void __launch_bounds__(256) k_nms64_resolve(const uint64_t *adj, const uint32_t *order, int n,
                                                       int words, uint32_t *keep_flag)
{
    __shared__ uint64_t removed[kNms64Max / 64];
    const int tid = threadIdx.x;
    for (int w = tid; w < words; w += 256) removed[w] = 0;
    for (int i = tid; i < n; i += 256) keep_flag[i] = 0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const int wi = i >> 6;
        const bool gone = (removed[wi] >> (i & 63)) & 1ull;               // uniform: LDS broadcast
        if (gone) continue;
        if (tid == 0) keep_flag[order[i]] = 1;
        __syncthreads();                                                   // everyone has read removed[wi]
        for (int w = wi + tid; w < words; w += 256) removed[w] |= adj[(size_t)i * words + w];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(1024) k_nms64_emit(const uint32_t *keep_flag, int n, int32_t *keep, int32_t *count)
{
    __shared__ int32_t s_wave[16];
    __shared__ int32_t s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const int f = (i < n && keep_flag[i]) ? 1 : 0;
        const uint64_t m = __ballot(f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; ++w) off += s_wave[w];
        if (f) keep[off + before] = i;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += s_wave[w]; s_base += t; }
        __syncthreads();
    }
    if (tid == 0) *count = s_base;
}

struct Nms64Layout { size_t keys_in, keys_out, vals_in, vals_out, adj, flags, sort_tmp, total; size_t sort_bytes; };

static int nms64_layout(int n, Nms64Layout &w)
{
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t N = (size_t)n, words = (N + 63) / 64;
    size_t o = 0;
    w.keys_in = o; o = up(o + N * 8);
    w.keys_out = o; o = up(o + N * 8);
    w.vals_in = o; o = up(o + N * 4);
    w.vals_out = o; o = up(o + N * 4);
    w.adj = o; o = up(o + N * words * 8);
    w.flags = o; o = up(o + N * 4);
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, need, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                                  (uint32_t *)nullptr, (uint32_t *)nullptr, N, 0, 64, (hipStream_t)0);
    if (e != hipSuccess) return (int)e;
    w.sort_bytes = need;
    w.sort_tmp = o; o = up(o + need);
    w.total = o;
    return 0;
}

}  // namespace ia

extern "C" size_t ia_nms_f64_workspace_bytes(int n)
{
    if (n < 1 || n > ia::kNms64Max) return 0;
    ia::Nms64Layout w;
    return ia::nms64_layout(n, w) ? 0 : w.total;
}

extern "C" int ia_nms_f64(const double *dets, int n, float iou_thr, int32_t *keep, int32_t *count,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || !count) return IA_E_ARG;
    if (n == 0) return ia::hip_status(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    if (n > ia::kNms64Max) return IA_E_LIMIT_BOXES;
    if (!dets || !keep || !workspace) return IA_E_ARG;
    ia::Nms64Layout w;
    int rc = ia::nms64_layout(n, w);
    if (rc) return rc;
    if (workspace_bytes < w.total) return IA_E_WORKSPACE;
    char *ws = static_cast<char *>(workspace);
    uint64_t *keys_in = (uint64_t *)(ws + w.keys_in), *keys_out = (uint64_t *)(ws + w.keys_out);
    uint32_t *vals_in = (uint32_t *)(ws + w.vals_in), *vals_out = (uint32_t *)(ws + w.vals_out);
    uint64_t *adj = (uint64_t *)(ws + w.adj);
    uint32_t *flags = (uint32_t *)(ws + w.flags);
    const int words = (n + 63) / 64;
    hipLaunchKernelGGL(ia::k_nms64_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dets, n, keys_in, vals_in);
    size_t need = w.sort_bytes;
    hipError_t e = rocprim::radix_sort_pairs_desc(ws + w.sort_tmp, need, keys_in, keys_out, vals_in, vals_out,
                                                  (size_t)n, 0, 64, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(ia::k_nms64_adj, dim3((unsigned)words, (unsigned)words), dim3(64), 0, s, dets, vals_out, n,
                       (double)iou_thr, words, adj);
    hipLaunchKernelGGL(ia::k_nms64_resolve, dim3(1), dim3(256), 0, s, adj, vals_out, n, words, flags);
    hipLaunchKernelGGL(ia::k_nms64_emit, dim3(1), dim3(1024), 0, s, flags, n, keep, count);
    return ia::hip_status(hipGetLastError());
}
