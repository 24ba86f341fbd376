// Small training-side kernels of the convolution autograd nodes (iouaware/train_fuse.py,
// iouaware/winograd_train.py; BASELINE config 5, the R-50 training iteration):
//
//   k_wino_weight        U[6a+b][i][o] = (G w[o][i] G^T)[a][b]      weights -> Winograd domain, every
//                        iteration (the weights change), forward and -- flipped, in/out swapped --
//                        for the input gradient.  Was a float64 einsum = a batched DGEMM per
//                        convolution (5.6 ms of a 48.7 ms iteration).
//   k_wino_weight_grad   dW[o][i] = G^T dU[.][i][o] G               the adjoint, after the
//                        Winograd-domain weight-gradient GEMM.
//   k_relu_bwd_colsum    g = dy * (y > 0),  db[c] = sum_rows g[row][c]   ReLU backward and the bias
//                        gradient of a channels-last activation in ONE pass (eager: a
//                        threshold_backward pass, then a column-sum reduction pass).
//
// All three are HBM-bound streaming kernels; the weight transforms move 9 + 36 floats per
// (in, out) pair (<= 47 MB for 512 x 512), the ReLU/bias pass 3 floats per element.
// G is F(4x4,3x3)'s kernel transform, in double on the device (G's entries are not fp32
// numbers; the products are rounded to fp32 once, like the float64 einsum they replace).
#include "ia_internal.hpp"

namespace ia {

struct WinoWeightArgs {
    const float *w; float *u;
    int n_in, n_out;
    long s_in, s_out, s_ky, s_kx;      // element strides of w for (in, out, ky, kx)
    int flip;
    double G[6][3];
};

__global__ __launch_bounds__(256) void k_wino_weight(WinoWeightArgs a)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.n_in * a.n_out;
    if (idx >= total) return;
    const int o = (int)(idx % a.n_out), i = (int)(idx / a.n_out);
    const float *p = a.w + (long)i * a.s_in + (long)o * a.s_out;
    double g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = a.flip ? 2 - ky : ky, xx = a.flip ? 2 - kx : kx;
            g[ky][kx] = (double)p[yy * a.s_ky + xx * a.s_kx];
        }
    double t[6][3];                                   // G g
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            t[r][c] = a.G[r][0] * g[0][c] + a.G[r][1] * g[1][c] + a.G[r][2] * g[2][c];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double v = t[r][0] * a.G[c][0] + t[r][1] * a.G[c][1] + t[r][2] * a.G[c][2];
            a.u[(long)(r * 6 + c) * total + idx] = (float)v;        // (36, n_in, n_out)
        }
}

struct WinoWeightGradArgs {
    const float *du; float *dw;        // du (36, n_in, n_out); dw (n_out, n_in, 3, 3), any strides
    int n_in, n_out;
    long s_in, s_out, s_ky, s_kx;
    double G[6][3];
};

__global__ __launch_bounds__(256) void k_wino_weight_grad(WinoWeightGradArgs a)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.n_in * a.n_out;
    if (idx >= total) return;
    const int o = (int)(idx % a.n_out), i = (int)(idx / a.n_out);
    double t[3][6];                                   // G^T dU
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) t[r][c] = 0.0;
    // the 36 loads first: next to their use they compile to load + s_waitcnt vmcnt(0) each,
    // 36 dependent round trips (17 us per launch for a 256 x 256 weight)
    float du[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) du[k] = a.du[(long)k * total + idx];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double v = (double)du[k * 6 + c];
#pragma unroll
            for (int r = 0; r < 3; ++r) t[r][c] += a.G[k][r] * v;
        }
    float *q = a.dw + (long)o * a.s_out + (long)i * a.s_in;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) v += t[r][k] * a.G[k][c];
            q[r * a.s_ky + c * a.s_kx] = (float)v;
        }
}

// rows x n (n % 4 == 0) row-major.  A workgroup of 256 threads = 16 column quads (64 columns, 256
// contiguous bytes of a row) x 16 row lanes walks one of S row strips of one column block.  Its
// column sums go to partial[cb][s][64]; k_colsum_finish adds the S partial rows of a block in a
// fixed order -- no atomics (a first version with one atomicAdd per column and workgroup spent
// 88 us per launch on ~1M colliding atomics; a last-workgroup-finishes variant paid 140 us for
// its device-scope fences = L2 write-backs under a streaming store), same bits every run.
struct ReluColsumArgs {
    const float *dy, *y; float *g, *db;
    float *partial;
    long rows; int n, S; long strip;
};

__global__ __launch_bounds__(256) void k_relu_bwd_colsum(ReluColsumArgs a)
{
    __shared__ float4 red[256];
    const int t = threadIdx.x, ql = t & 15, rl = t >> 4;
    const int cb = blockIdx.y, s = blockIdx.x;
    const int q = cb * 16 + ql, qn = a.n >> 2;
    const long r0 = (long)s * a.strip;
    const long r1 = r0 + a.strip < a.rows ? r0 + a.strip : a.rows;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < qn) {
        long r = r0 + rl;
        for (; r + 48 < r1; r += 64) {                 // four rows in flight per thread
            float4 d[4], yy[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                d[u] = *reinterpret_cast<const float4 *>(a.dy + (r + 16 * u) * a.n + 4L * q);
            if (a.y) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    yy[u] = *reinterpret_cast<const float4 *>(a.y + (r + 16 * u) * a.n + 4L * q);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    d[u].x = yy[u].x > 0.f ? d[u].x : 0.f; d[u].y = yy[u].y > 0.f ? d[u].y : 0.f;
                    d[u].z = yy[u].z > 0.f ? d[u].z : 0.f; d[u].w = yy[u].w > 0.f ? d[u].w : 0.f;
                    *reinterpret_cast<float4 *>(a.g + (r + 16 * u) * a.n + 4L * q) = d[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc.x += d[u].x; acc.y += d[u].y; acc.z += d[u].z; acc.w += d[u].w;
            }
        }
        for (; r < r1; r += 16) {
            float4 d = *reinterpret_cast<const float4 *>(a.dy + r * a.n + 4L * q);
            if (a.y) {
                const float4 yy = *reinterpret_cast<const float4 *>(a.y + r * a.n + 4L * q);
                d.x = yy.x > 0.f ? d.x : 0.f; d.y = yy.y > 0.f ? d.y : 0.f;
                d.z = yy.z > 0.f ? d.z : 0.f; d.w = yy.w > 0.f ? d.w : 0.f;
                *reinterpret_cast<float4 *>(a.g + r * a.n + 4L * q) = d;
            }
            acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
        }
    }
    if (!a.db) return;
    red[t] = acc;
    __syncthreads();
    if (t < 16) {
        float4 v = red[t];
        for (int k = 1; k < 16; ++k) {
            const float4 w = red[k * 16 + t];
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        *reinterpret_cast<float4 *>(a.partial + ((long)cb * a.S + s) * 64 + 4 * t) = v;
    }
}

// db[c] = sum over the S partial rows of column block c / 64, in a fixed order
__global__ __launch_bounds__(256) void k_colsum_finish(ReluColsumArgs a)
{
    __shared__ float4 red[256];
    const int t = threadIdx.x, ql = t & 15, rl = t >> 4;
    const int cb = blockIdx.x, q = cb * 16 + ql, qn = a.n >> 2;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight partial rows requested at a time (clamped addresses), added in the same fixed order:
    // one load per iteration compiled to load + s_waitcnt, up to 32 dependent round trips
    for (int k0 = rl; k0 < a.S; k0 += 16 * 8) {
        float4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 16 * u;
            w[u] = *reinterpret_cast<const float4 *>(a.partial + ((long)cb * a.S + (k < a.S ? k : a.S - 1)) * 64 + 4 * ql);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + 16 * u < a.S) { v.x += w[u].x; v.y += w[u].y; v.z += w[u].z; v.w += w[u].w; }
    }
    red[t] = v;
    __syncthreads();
    if (t < 16 && q < qn) {
        float4 o = red[t];
        for (int k = 1; k < 16; ++k) {
            const float4 w = red[k * 16 + t];
            o.x += w.x; o.y += w.y; o.z += w.z; o.w += w.w;
        }
        *reinterpret_cast<float4 *>(a.db + 4 * q) = o;
    }
}

// Eval-mode BatchNorm folded into the convolution in front of it, per output channel o (one
// workgroup each, K = Cin*kh*kw weights that are dense in memory whatever the memory format):
//   forward    s = gamma * inv_std,  w'[o][:] = w[o][:] * s,  b'[o] = beta - mean * s
//   backward   dw[o][:] = dw'[o][:] * s,  dgamma = inv_std * (sum_j dw'[o][j] w[o][j] - mean * db'),
//              dbeta = db'
// (eager: 3 small kernels forward and ~8 backward per convolution, 42 convolutions per iteration)
struct FoldArgs {
    const float *w, *gamma, *beta, *mean, *inv;
    const float *dwp, *dbp;            // backward: gradients w.r.t. w', b' (dbp may be NULL = 0)
    float *w_out, *b_out;              // forward outputs
    float *dw, *dgamma, *dbeta;        // backward outputs (dw may be NULL: frozen weight)
    int K;
};

__global__ __launch_bounds__(256) void k_bn_fold_fwd(FoldArgs a)
{
    const int o = blockIdx.x;
    const float s = a.gamma[o] * a.inv[o];
    const float *w = a.w + (long)o * a.K;
    float *q = a.w_out + (long)o * a.K;
    for (int j = threadIdx.x; j < a.K; j += 256) q[j] = w[j] * s;
    if (threadIdx.x == 0) a.b_out[o] = a.beta[o] - a.mean[o] * s;
}

__global__ __launch_bounds__(256) void k_bn_fold_bwd(FoldArgs a)
{
    __shared__ float red[4];
    const int o = blockIdx.x;
    const float s = a.gamma[o] * a.inv[o];
    const float *w = a.w + (long)o * a.K, *g = a.dwp + (long)o * a.K;
    float *q = a.dw ? a.dw + (long)o * a.K : nullptr;
    float acc = 0.f;
    for (int j = threadIdx.x; j < a.K; j += 256) {
        const float gv = g[j];
        acc += gv * w[j];
        if (q) q[j] = gv * s;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float dot = (red[0] + red[1]) + (red[2] + red[3]);
        const float db = a.dbp ? a.dbp[o] : 0.f;
        a.dgamma[o] = a.inv[o] * (dot - a.mean[o] * db);
        a.dbeta[o] = db;
    }
}

}  // namespace ia

extern "C" {

int ia_wino_weight_transform(const float *w, int n_in, int n_out, int64_t stride_in,
                             int64_t stride_out, int64_t stride_ky, int64_t stride_kx, int flip,
                             const double *G, float *U, void *stream)
{
    if (!w || !U || !G || n_in < 1 || n_out < 1) return IA_E_ARG;
    ia::WinoWeightArgs a;
    a.w = w; a.u = U; a.n_in = n_in; a.n_out = n_out;
    a.s_in = stride_in; a.s_out = stride_out; a.s_ky = stride_ky; a.s_kx = stride_kx;
    a.flip = flip ? 1 : 0;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 3; ++c) a.G[r][c] = G[r * 3 + c];
    const long total = (long)n_in * n_out;
    hipLaunchKernelGGL(ia::k_wino_weight, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}

int ia_wino_weight_grad(const float *dU, int n_in, int n_out, const double *G, float *dW,
                        int64_t stride_in, int64_t stride_out, int64_t stride_ky, int64_t stride_kx,
                        void *stream)
{
    if (!dU || !dW || !G || n_in < 1 || n_out < 1) return IA_E_ARG;
    ia::WinoWeightGradArgs a;
    a.du = dU; a.dw = dW; a.n_in = n_in; a.n_out = n_out;
    a.s_in = stride_in; a.s_out = stride_out; a.s_ky = stride_ky; a.s_kx = stride_kx;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 3; ++c) a.G[r][c] = G[r * 3 + c];
    const long total = (long)n_in * n_out;
    hipLaunchKernelGGL(ia::k_wino_weight_grad, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}

size_t ia_relu_bwd_bias_grad_workspace_bytes(int64_t rows, int n)
{
    if (rows < 1 || n < 4 || (n & 3) || n > 65536) return 0;
    const long ncb = (n / 4 + 15) / 16;
    return (size_t)ncb * IA_COLSUM_MAX_STRIPS * 64 * sizeof(float);
}

int ia_relu_bwd_bias_grad(const float *dy, const float *y, int64_t rows, int n, float *g, float *db,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    if (!dy || rows < 1 || n < 4 || (n & 3) || n > 65536 || (y && !g) || (!y && !db)) return IA_E_ARG;
    if (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)g | (uintptr_t)db | (uintptr_t)workspace) & 15u)
        return IA_E_ARG;
    const int ncb = (n / 4 + 15) / 16;
    if (db && (!workspace ||
               workspace_bytes < ia_relu_bwd_bias_grad_workspace_bytes(rows, n)))
        return IA_E_WORKSPACE;
    // ~2048 workgroups in all, at least 64 rows (four per row lane) per strip
    long S = 2048 / ncb;
    if (S > IA_COLSUM_MAX_STRIPS) S = IA_COLSUM_MAX_STRIPS;
    if (S > (rows + 63) / 64) S = (rows + 63) / 64;
    if (S < 1) S = 1;
    long strip = (rows + S - 1) / S;
    strip = (strip + 15) / 16 * 16;
    S = (rows + strip - 1) / strip;
    ia::ReluColsumArgs a;
    a.dy = dy; a.y = y; a.g = g; a.db = db; a.partial = (float *)workspace;
    a.rows = rows; a.n = n; a.S = (int)S; a.strip = strip;
    hipLaunchKernelGGL(ia::k_relu_bwd_colsum, dim3((unsigned)S, (unsigned)ncb), dim3(256), 0,
                       (hipStream_t)stream, a);
    if (db)
        hipLaunchKernelGGL(ia::k_colsum_finish, dim3((unsigned)ncb), dim3(256), 0,
                           (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}

}  // extern "C"

extern "C" int ia_bn_fold_fwd(const float *w, const float *gamma, const float *beta, const float *mean,
                              const float *inv_std, int cout, int K, float *w_out, float *b_out,
                              void *stream)
{
    if (!w || !gamma || !beta || !mean || !inv_std || !w_out || !b_out || cout < 1 || K < 1)
        return IA_E_ARG;
    ia::FoldArgs a = {};
    a.w = w; a.gamma = gamma; a.beta = beta; a.mean = mean; a.inv = inv_std;
    a.w_out = w_out; a.b_out = b_out; a.K = K;
    hipLaunchKernelGGL(ia::k_bn_fold_fwd, dim3((unsigned)cout), dim3(256), 0, (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}

extern "C" int ia_bn_fold_bwd(const float *dw_folded, const float *db_folded, const float *w,
                              const float *gamma, const float *mean, const float *inv_std, int cout,
                              int K, float *dw, float *dgamma, float *dbeta, void *stream)
{
    if (!dw_folded || !w || !gamma || !mean || !inv_std || !dgamma || !dbeta || cout < 1 || K < 1)
        return IA_E_ARG;
    ia::FoldArgs a = {};
    a.w = w; a.gamma = gamma; a.mean = mean; a.inv = inv_std; a.dwp = dw_folded; a.dbp = db_folded;
    a.dw = dw; a.dgamma = dgamma; a.dbeta = dbeta; a.K = K;
    hipLaunchKernelGGL(ia::k_bn_fold_bwd, dim3((unsigned)cout), dim3(256), 0, (hipStream_t)stream, a);
    return ia::hip_status(hipGetLastError());
}
