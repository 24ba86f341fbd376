// 1x1 convolutions of the backbone as plain GEMMs on the channels-last activation, with the folded
// BatchNorm bias, the residual and the ReLU inside the library GEMM's epilogue (hipBLASLt):
//   D (rows x n) = act(A (rows x k) . W (k x n) + bias[n] + residual (rows x n)),  rows = B*H*W
// replaces  conv1x1 -> bn -> (+identity) -> relu  of Bottleneck.forward (reference
// mmdet/models/backbones/resnet.py:215-255) at inference: no separate elementwise pass over the
// 4*planes-channel activation.  A plain library GEMM: nothing here is hand-written math; the file
// only owns the handle, the per-shape algorithm choice (the first call of a shape times the
// heuristic's top candidates on the caller's stream) and the row-major <-> column-major mapping
//   D^T (n x rows) = W^T (n x k) . A^T (k x rows)      (all "N" operands in column-major terms).
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <mutex>
#include <tuple>
#include "ia_internal.hpp"

namespace ia {

struct LtState {
    // one library handle per stream: a handle owns device-side argument buffers, so launches that
    // may overlap on different streams must not share one
    std::map<hipStream_t, hipblasLtHandle_t> handles;
    std::mutex mu;
    std::map<std::tuple<int64_t, int, int, int, int, int>, hipblasLtMatmulAlgo_t> algos;
};

static LtState &lt_state()
{
    static LtState s;
    return s;
}

static int lt_status(hipblasStatus_t s) { return s == HIPBLAS_STATUS_SUCCESS ? 0 : 2000 + (int)s; }

// batch > 1: strided batched, A / W / D advance by rows*k / k*n / rows*n per matrix
// dtype: IA_F32, or IA_BF16 (A / W / residual / D bf16, bias fp32, fp32 accumulation)
static int lt_matmul(const void *A, const void *W, const float *bias, const void *residual,
                     void *D, int64_t rows, int k, int n, int relu, int batch, int dtype,
                     void *workspace, size_t workspace_bytes, void *stream)
{
    if (dtype != IA_F32 && dtype != IA_BF16) return IA_E_ARG;
    const hipDataType dt = (dtype == IA_F32) ? HIP_R_32F : HIP_R_16BF;
    if (!A || !W || !D || rows < 1 || k < 1 || n < 1 || batch < 1 || residual == D || A == D)
        return IA_E_ARG;
    if (workspace_bytes && !workspace) return IA_E_ARG;
    ia::LtState &st = ia::lt_state();
    std::lock_guard<std::mutex> lock(st.mu);
    int rc;
    hipStream_t s = (hipStream_t)stream;
    hipblasLtHandle_t &handle = st.handles[s];
    if (!handle && (rc = ia::lt_status(hipblasLtCreate(&handle)))) return rc;

    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    hipblasLtMatmulPreference_t pref = nullptr;
    auto cleanup = [&]() {
        if (pref) hipblasLtMatmulPreferenceDestroy(pref);
        if (la) hipblasLtMatrixLayoutDestroy(la);
        if (lb) hipblasLtMatrixLayoutDestroy(lb);
        if (lc) hipblasLtMatrixLayoutDestroy(lc);
        if (desc) hipblasLtMatmulDescDestroy(desc);
    };
#define IA_LT(x) do { if ((rc = ia::lt_status(x))) { cleanup(); return rc; } } while (0)
    IA_LT(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t opn = HIPBLAS_OP_N;
    IA_LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opn, sizeof(opn)));
    IA_LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opn, sizeof(opn)));
    hipblasLtEpilogue_t ep = bias ? (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS)
                                  : (relu ? HIPBLASLT_EPILOGUE_RELU : HIPBLASLT_EPILOGUE_DEFAULT);
    IA_LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
    if (bias) {
        IA_LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
        const hipDataType bt = HIP_R_32F;
        IA_LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    }
    IA_LT(hipblasLtMatrixLayoutCreate(&la, dt, (uint64_t)n, (uint64_t)k, (int64_t)n));
    IA_LT(hipblasLtMatrixLayoutCreate(&lb, dt, (uint64_t)k, (uint64_t)rows, (int64_t)k));
    IA_LT(hipblasLtMatrixLayoutCreate(&lc, dt, (uint64_t)n, (uint64_t)rows, (int64_t)n));
    if (batch > 1) {
        const int32_t bc = batch;
        const int64_t sa = (int64_t)k * n, sb = rows * k, sc = rows * n;
        hipblasLtMatrixLayout_t ls[3] = {la, lb, lc};
        const int64_t so[3] = {sa, sb, sc};
        for (int i = 0; i < 3; ++i) {
            IA_LT(hipblasLtMatrixLayoutSetAttribute(ls[i], HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)));
            IA_LT(hipblasLtMatrixLayoutSetAttribute(ls[i], HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET,
                                                    &so[i], sizeof(so[i])));
        }
    }
    const float alpha = 1.0f, beta = residual ? 1.0f : 0.0f;
    const void *C = residual ? residual : D;
    const int flags = (bias ? 1 : 0) | (relu ? 2 : 0) | (residual ? 4 : 0);
    const auto key = std::make_tuple(rows, k, n, flags, batch, dtype);
    auto it = st.algos.find(key);
    if (it == st.algos.end()) {
        IA_LT(hipblasLtMatmulPreferenceCreate(&pref));
        IA_LT(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES,
                                                    &workspace_bytes, sizeof(workspace_bytes)));
        hipblasLtMatmulHeuristicResult_t res[16];
        int nres = 0;
        IA_LT(hipblasLtMatmulAlgoGetHeuristic(handle, desc, la, lb, lc, lc, pref, 16, res, &nres));
        if (nres < 1) { cleanup(); return IA_E_ARG; }
        // first call of this shape: time the candidates on the caller's stream (the output is
        // simply rewritten; C != D, so every run computes the same result)
        int best = 0;
        float best_ms = 1e30f;
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            for (int a = 0; a < nres; ++a) {
                bool ok = true;
                for (int r = 0; r < 2 && ok; ++r)
                    ok = hipblasLtMatmul(handle, desc, &alpha, W, la, A, lb, &beta, C, lc, D, lc,
                                         &res[a].algo, workspace, workspace_bytes, s) == HIPBLAS_STATUS_SUCCESS;
                if (!ok) continue;
                (void)hipEventRecord(e0, s);
                for (int r = 0; r < 4; ++r)
                    hipblasLtMatmul(handle, desc, &alpha, W, la, A, lb, &beta, C, lc, D, lc,
                                    &res[a].algo, workspace, workspace_bytes, s);
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best_ms) { best_ms = ms; best = a; }
            }
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        it = st.algos.emplace(key, res[best].algo).first;
    }
    rc = ia::lt_status(hipblasLtMatmul(handle, desc, &alpha, W, la, A, lb, &beta, C, lc, D, lc,
                                       &it->second, workspace, workspace_bytes, s));
#undef IA_LT
    cleanup();
    return rc;
}

}  // namespace ia

extern "C" int ia_linear_bias_act(const float *A, const float *W, const float *bias,
                                  const float *residual, float *D, int64_t rows, int k, int n,
                                  int relu, void *workspace, size_t workspace_bytes, void *stream)
{
    return ia::lt_matmul(A, W, bias, residual, D, rows, k, n, relu, 1, IA_F32, workspace,
                         workspace_bytes, stream);
}

extern "C" int ia_linear_bias_act_bf16(const void *A, const void *W, const float *bias,
                                       const void *residual, void *D, int64_t rows, int k, int n,
                                       int relu, void *workspace, size_t workspace_bytes,
                                       void *stream)
{
    return ia::lt_matmul(A, W, bias, residual, D, rows, k, n, relu, 1, IA_BF16, workspace,
                         workspace_bytes, stream);
}

extern "C" int ia_batched_gemm(const float *A, const float *W, float *D, int batch, int64_t rows,
                               int k, int n, void *workspace, size_t workspace_bytes, void *stream)
{
    return ia::lt_matmul(A, W, nullptr, nullptr, D, rows, k, n, 0, batch, IA_F32, workspace,
                         workspace_bytes, stream);
}
