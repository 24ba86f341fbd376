// 1x1 convolutions of the backbone as plain GEMMs on the channels-last activation, with the folded
// BatchNorm bias, the residual and the ReLU inside the library GEMM's epilogue (hipBLASLt):
//   D (rows x n) = act(A (rows x k) . W (k x n) + bias[n] + residual (rows x n)),  rows = B*H*W
// replaces  conv1x1 -> bn -> (+identity) -> relu  of Bottleneck.forward (reference
// mmdet/models/backbones/resnet.py:215-255) at inference: no separate elementwise pass over the
// 4*planes-channel activation.  A plain library GEMM: nothing here is hand-written math; the file
// only owns the handle, the per-shape algorithm choice (the first call of a shape times the
// heuristic's top candidates on the caller's stream) and the row-major <-> column-major mapping
//   D^T (n x rows) = W^T (n x k) . A^T (k x rows)      (all "N" operands in column-major terms).
// Training (iouaware/train_fuse.py, winograd_train.py) adds the weight stored (n, k) -- read through
// the library's transpose flag, no transposed copy per iteration -- and the weight-gradient product
// G^T X, a reduction over the pixel rows with a small result, for which the timed candidates include
// the library's split-K kernels.
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
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
    typedef std::tuple<int64_t, int64_t, int64_t, int, int, int> Key;     // m, n, k, flags, batch, dtype
    std::map<Key, hipblasLtMatmulAlgo_t> algos;
    // How the kernel of a new shape is chosen.  2 (the default) = FROZEN: the entry of the tuning
    // table (ia_gemm_table_add: library solution indices found offline by tools/tune_gemm.py and
    // committed with the package), else the library heuristic's first result -- nothing is timed, so
    // the same process on the same library computes the same bits in every run.  0 / 1 are the
    // OFFLINE modes that write the table: time the heuristic's top 16 / every kernel of the library
    // that supports the problem (~250 for fp32: ~0.3 s per shape) at the first call of a shape.
    int tuning = -1;
    std::map<Key, int> table;          // shape -> library solution index (frozen mode)
    std::map<Key, int> chosen;         // shape -> solution index in use (ia_gemm_table_dump)
    long table_hits = 0, table_misses = 0, table_stale = 0;
};

static LtState &lt_state()
{
    static LtState s;
    return s;
}

static int lt_status(hipblasStatus_t s) { return s == HIPBLAS_STATUS_SUCCESS ? 0 : 2000 + (int)s; }

// Column-major BLAS semantics: D (m x n) = act(op(A) (m x k) . op(B) (k x n) + bias[m] + C), all
// matrices column-major with the given leading dimensions; batch > 1: strided batched with the
// element strides sa / sb / sd.  dtype: IA_F32, or IA_BF16 (A / B / C / D bf16, bias fp32, fp32
// accumulation).  `shape_tag` separates the callers in the algorithm cache.
static int lt_gemm(int transa, int transb, int64_t m, int64_t n, int64_t k, const void *A, int64_t lda,
                   int64_t sa, const void *B, int64_t ldb, int64_t sb, const float *bias,
                   const void *residual, void *D, int64_t ldd, int64_t sd, int relu, int batch,
                   int dtype, void *workspace, size_t workspace_bytes, void *stream, int tag = 0)
{
    if (dtype != IA_F32 && dtype != IA_BF16) return IA_E_ARG;
    const hipDataType dt = (dtype == IA_F32) ? HIP_R_32F : HIP_R_16BF;
    if (!A || !B || !D || m < 1 || n < 1 || k < 1 || batch < 1 || residual == D || A == D || B == D)
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
    const hipblasOperation_t opa = transa ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    const hipblasOperation_t opb = transb ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    IA_LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa)));
    IA_LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb)));
    hipblasLtEpilogue_t ep = bias ? (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS)
                                  : (relu ? HIPBLASLT_EPILOGUE_RELU : HIPBLASLT_EPILOGUE_DEFAULT);
    IA_LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
    if (bias) {
        IA_LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
        const hipDataType bt = HIP_R_32F;
        IA_LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    }
    IA_LT(hipblasLtMatrixLayoutCreate(&la, dt, (uint64_t)(transa ? k : m), (uint64_t)(transa ? m : k), lda));
    IA_LT(hipblasLtMatrixLayoutCreate(&lb, dt, (uint64_t)(transb ? n : k), (uint64_t)(transb ? k : n), ldb));
    IA_LT(hipblasLtMatrixLayoutCreate(&lc, dt, (uint64_t)m, (uint64_t)n, ldd));
    if (batch > 1) {
        const int32_t bc = batch;
        hipblasLtMatrixLayout_t ls[3] = {la, lb, lc};
        const int64_t so[3] = {sa, sb, sd};
        for (int i = 0; i < 3; ++i) {
            IA_LT(hipblasLtMatrixLayoutSetAttribute(ls[i], HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)));
            IA_LT(hipblasLtMatrixLayoutSetAttribute(ls[i], HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET,
                                                    &so[i], sizeof(so[i])));
        }
    }
    const float alpha = 1.0f, beta = residual ? 1.0f : 0.0f;
    const void *C = residual ? residual : D;
    const int flags = (bias ? 1 : 0) | (relu ? 2 : 0) | (residual ? 4 : 0) | (transa ? 8 : 0) |
                      (transb ? 16 : 0) | (tag << 5);
    const auto key = std::make_tuple(m, n, k, flags, batch, dtype);
    auto it = st.algos.find(key);
    if (st.tuning < 0) {
        const char *tune = getenv("IA_GEMM_TUNE");
        st.tuning = (tune && !strcmp(tune, "all")) ? 1 : (tune && !strcmp(tune, "heuristic")) ? 0 : 2;
    }
    if (it == st.algos.end() && st.tuning == 2) {
        // frozen: the committed table's solution when the library still has it for this problem
        auto te = st.table.find(key);
        if (te != st.table.end()) {
            std::vector<int> idx(1, te->second);
            std::vector<hipblasLtMatmulHeuristicResult_t> one;
            size_t need = 0;
            if (hipblaslt_ext::getAlgosFromIndex(handle, idx, one) == HIPBLAS_STATUS_SUCCESS && one.size() == 1 &&
                hipblaslt_ext::matmulIsAlgoSupported(handle, desc, &alpha, la, lb, &beta, lc, lc, one[0].algo,
                                                     need) == HIPBLAS_STATUS_SUCCESS &&
                need <= workspace_bytes) {
                it = st.algos.emplace(key, one[0].algo).first;
                st.chosen[key] = te->second;
                ++st.table_hits;
            } else {
                ++st.table_stale;
            }
        }
        if (it == st.algos.end()) {
            IA_LT(hipblasLtMatmulPreferenceCreate(&pref));
            IA_LT(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES,
                                                        &workspace_bytes, sizeof(workspace_bytes)));
            hipblasLtMatmulHeuristicResult_t first;
            int nres = 0;
            IA_LT(hipblasLtMatmulAlgoGetHeuristic(handle, desc, la, lb, lc, lc, pref, 1, &first, &nres));
            if (nres < 1) { cleanup(); return IA_E_ARG; }
            it = st.algos.emplace(key, first.algo).first;
            st.chosen[key] = hipblaslt_ext::getIndexFromAlgo(first.algo);
            ++st.table_misses;
        }
    }
    if (it == st.algos.end()) {
        IA_LT(hipblasLtMatmulPreferenceCreate(&pref));
        IA_LT(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES,
                                                    &workspace_bytes, sizeof(workspace_bytes)));
        std::vector<hipblasLtMatmulHeuristicResult_t> res(16);
        int nres = 0;
        IA_LT(hipblasLtMatmulAlgoGetHeuristic(handle, desc, la, lb, lc, lc, pref, 16, res.data(), &nres));
        res.resize(nres > 0 ? nres : 0);
        if (st.tuning == 1) {
            // every kernel of the library that supports the problem, not only the heuristic's top 16
            std::vector<hipblasLtMatmulHeuristicResult_t> all;
            if (hipblaslt_ext::getAllAlgos(handle, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, opa, opb, dt,
                                           dt, dt, dt, HIPBLAS_COMPUTE_32F, all) == HIPBLAS_STATUS_SUCCESS)
                for (auto &r : all) {
                    size_t need = 0;
                    if (hipblaslt_ext::matmulIsAlgoSupported(handle, desc, &alpha, la, lb, &beta, lc, lc,
                                                             r.algo, need) == HIPBLAS_STATUS_SUCCESS &&
                        need <= workspace_bytes)
                        res.push_back(r);
                }
        }
        nres = (int)res.size();
        if (nres < 1) { cleanup(); return IA_E_ARG; }
        // first call of this shape: time the candidates on the caller's stream (the output is
        // simply rewritten; C != D, so every run computes the same result).  Two passes: one timed
        // run each, then the best eight again with four runs.
        int best = 0;
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            auto time_one = [&](int a, int runs) -> float {
                if (hipblasLtMatmul(handle, desc, &alpha, A, la, B, lb, &beta, C, lc, D, lc, &res[a].algo,
                                    workspace, workspace_bytes, s) != HIPBLAS_STATUS_SUCCESS)
                    return 1e30f;
                (void)hipEventRecord(e0, s);
                for (int r = 0; r < runs; ++r)
                    hipblasLtMatmul(handle, desc, &alpha, A, la, B, lb, &beta, C, lc, D, lc, &res[a].algo,
                                    workspace, workspace_bytes, s);
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                return ms / runs;
            };
            std::vector<std::pair<float, int>> first;
            for (int a = 0; a < nres; ++a) first.emplace_back(time_one(a, nres > 16 ? 1 : 2), a);
            std::sort(first.begin(), first.end());
            float best_ms = 1e30f;
            for (size_t i = 0; i < first.size() && i < 8; ++i) {
                if (first[i].first >= 1e29f) break;
                // short kernels: enough runs for ~0.5 ms between the events (a 30 us kernel timed
                // over 4 runs is decided by launch jitter)
                int runs = first[0].first > 0.f ? (int)(0.5f / first[0].first) : 4;
                runs = runs < 4 ? 4 : (runs > 64 ? 64 : runs);
                const float ms = time_one(first[i].second, runs);
                if (ms < best_ms) { best_ms = ms; best = first[i].second; }
            }
            if (getenv("IA_GEMM_TUNE_VERBOSE"))
                fprintf(stderr, "[ia gemm] m=%ld n=%ld k=%ld flags=%d batch=%d: %d candidates, best #%d "
                        "%.1f us (heuristic first: %.1f us)\n", (long)m, (long)n, (long)k, flags, batch,
                        nres, best, best_ms * 1e3f, [&]() { for (auto &f : first) if (f.second == 0) return f.first * 1e3f; return 0.f; }());
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        it = st.algos.emplace(key, res[best].algo).first;
        st.chosen[key] = hipblaslt_ext::getIndexFromAlgo(res[best].algo);
    }
    rc = ia::lt_status(hipblasLtMatmul(handle, desc, &alpha, A, la, B, lb, &beta, C, lc, D, lc,
                                       &it->second, workspace, workspace_bytes, s));
#undef IA_LT
    cleanup();
    return rc;
}

// row-major view used by the 1x1 convolutions:  D (rows x n) = act(A (rows x k) . W + bias[n] + R),
// i.e. D^T (n x rows) = op(W) (n x k) . A^T (k x rows) in column-major terms.
//   w_nk == 0: W stored (k, n) row-major (the inference layout);  w_nk == 1: W stored (n, k)
//   row-major -- a convolution weight (Cout, Cin) as it is, read through the transpose flag
static int lt_matmul(const void *A, const void *W, const float *bias, const void *residual,
                     void *D, int64_t rows, int k, int n, int relu, int batch, int dtype,
                     void *workspace, size_t workspace_bytes, void *stream, int w_nk = 0)
{
    if (rows < 1 || k < 1 || n < 1) return IA_E_ARG;
    return lt_gemm(w_nk ? 1 : 0, 0, n, rows, k, W, w_nk ? k : n, (int64_t)k * n, A, k, rows * k, bias,
                   residual, D, n, rows * n, relu, batch, dtype, workspace, workspace_bytes, stream);
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

/* 1x1 convolution with a stride on a channels-last activation (the projection shortcut of the
 * first block of ResNet stages 2-4, resnet.py:436-449) as ONE strided-batched library GEMM -- no
 * gather copy: batch item (b, yo) = output row yo of image b; its Wo input pixels lie `stride`
 * pixels apart (leading dimension stride * k) and consecutive batch items stride * W * k apart
 * (images are contiguous and H % stride == 0, so the item index runs through the images).        */
extern "C" int ia_conv1x1_strided(const void *x, const void *W_kn, const float *bias, const void *residual,
                                  void *D, int B, int H, int W, int k, int n, int stride, int relu,
                                  int dtype, void *workspace, size_t workspace_bytes, void *stream)
{
    if (B < 1 || H < 1 || W < 1 || k < 1 || n < 1 || stride < 1) return IA_E_ARG;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const size_t esz = dtype == IA_F32 ? 4 : 2;
    if (H % stride == 0)
        return ia::lt_gemm(0, 0, n, Wo, k, W_kn, n, 0, x, (int64_t)stride * k, (int64_t)stride * W * k, bias,
                           residual, D, n, (int64_t)Wo * n, relu, B * Ho, dtype, workspace, workspace_bytes,
                           stream, 1);
    for (int b = 0; b < B; ++b) {          // rows of the next image do not continue the item stride
        const char *xb = (const char *)x + (size_t)b * H * W * k * esz;
        const char *rb = residual ? (const char *)residual + (size_t)b * Ho * Wo * n * esz : nullptr;
        char *db = (char *)D + (size_t)b * Ho * Wo * n * esz;
        int rc = ia::lt_gemm(0, 0, n, Wo, k, W_kn, n, 0, xb, (int64_t)stride * k, (int64_t)stride * W * k,
                             bias, rb, db, n, (int64_t)Wo * n, relu, Ho, dtype, workspace, workspace_bytes,
                             stream, 1);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int ia_batched_gemm(const float *A, const float *W, float *D, int batch, int64_t rows,
                               int k, int n, void *workspace, size_t workspace_bytes, void *stream)
{
    return ia::lt_matmul(A, W, nullptr, nullptr, D, rows, k, n, 0, batch, IA_F32, workspace,
                         workspace_bytes, stream);
}

/* training: see include/iouaware.h */
extern "C" int ia_linear_bias_act_wt(const float *A, const float *W_nk, const float *bias,
                                     const float *residual, float *D, int64_t rows, int k, int n,
                                     int relu, void *workspace, size_t workspace_bytes, void *stream)
{
    return ia::lt_matmul(A, W_nk, bias, residual, D, rows, k, n, relu, 1, IA_F32, workspace,
                         workspace_bytes, stream, 1);
}

extern "C" int ia_gemm_tn(const float *G, const float *X, float *D, int batch, int64_t rows, int n,
                          int k, void *workspace, size_t workspace_bytes, void *stream)
{
    // D (n x k) row-major = G^T (n x rows) . X (rows x k):  column-major D^T (k x n) = X_cm (k x rows)
    // . op_T(G_cm (n x rows))
    if (rows < 1 || k < 1 || n < 1) return IA_E_ARG;
    return ia::lt_gemm(0, 1, k, n, rows, X, k, rows * k, G, n, rows * n, nullptr, nullptr, D, k,
                       (int64_t)n * k, 0, batch, IA_F32, workspace, workspace_bytes, stream);
}

extern "C" int ia_gemm_tuning(int mode)
{
    ia::LtState &st = ia::lt_state();
    std::lock_guard<std::mutex> lock(st.mu);
    const int prev = st.tuning < 0 ? 2 : st.tuning;
    if (mode >= 0 && mode <= 2) st.tuning = mode;
    return prev;
}

/* ---- the tuning table (include/iouaware.h) ---- */
extern "C" int ia_gemm_table_clear(void)
{
    ia::LtState &st = ia::lt_state();
    std::lock_guard<std::mutex> lock(st.mu);
    st.table.clear();
    st.algos.clear();
    st.chosen.clear();
    st.table_hits = st.table_misses = st.table_stale = 0;
    return 0;
}

extern "C" int ia_gemm_table_add(int64_t m, int64_t n, int64_t k, int flags, int batch, int dtype,
                                 int solution_index)
{
    if (m < 1 || n < 1 || k < 1 || batch < 1 || solution_index < 0) return IA_E_ARG;
    ia::LtState &st = ia::lt_state();
    std::lock_guard<std::mutex> lock(st.mu);
    const auto key = std::make_tuple(m, n, k, flags, batch, dtype);
    st.table[key] = solution_index;
    st.algos.erase(key);               // a shape already resolved is resolved again from the table
    st.chosen.erase(key);
    return 0;
}

extern "C" int ia_gemm_table_dump(int64_t *rows7, int capacity)
{
    ia::LtState &st = ia::lt_state();
    std::lock_guard<std::mutex> lock(st.mu);
    int i = 0;
    for (auto &e : st.chosen) {
        if (rows7 && i < capacity) {
            int64_t *r = rows7 + (size_t)i * 7;
            r[0] = std::get<0>(e.first); r[1] = std::get<1>(e.first); r[2] = std::get<2>(e.first);
            r[3] = std::get<3>(e.first); r[4] = std::get<4>(e.first); r[5] = std::get<5>(e.first);
            r[6] = e.second;
        }
        ++i;
    }
    return i;
}

extern "C" int ia_gemm_table_stats(int64_t *hits_misses_stale)
{
    ia::LtState &st = ia::lt_state();
    std::lock_guard<std::mutex> lock(st.mu);
    if (!hits_misses_stale) return IA_E_ARG;
    hits_misses_stale[0] = st.table_hits;
    hits_misses_stale[1] = st.table_misses;
    hits_misses_stale[2] = st.table_stale;
    return 0;
}

extern "C" int ia_gemm_library_version(void)
{
    ia::LtState &st = ia::lt_state();
    std::lock_guard<std::mutex> lock(st.mu);
    hipblasLtHandle_t h = nullptr;
    for (auto &e : st.handles) if (e.second) { h = e.second; break; }
    bool own = false;
    if (!h) { if (hipblasLtCreate(&h) != HIPBLAS_STATUS_SUCCESS) return -1; own = true; }
    int v = -1;
    if (hipblasLtGetVersion(h, &v) != HIPBLAS_STATUS_SUCCESS) v = -1;
    if (own) hipblasLtDestroy(h);
    return v;
}
