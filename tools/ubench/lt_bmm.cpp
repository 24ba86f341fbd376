// Micro-benchmark (not product code): the head's Winograd batched GEMM through hipBLASLt directly,
// all heuristic candidates timed:  M[b] (T x N) = V[b] (T x K) . U[b] (K x N), row-major, b < batch.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define CB(x) do { hipblasStatus_t s = (x); if (s != HIPBLAS_STATUS_SUCCESS) { printf("hipblaslt error %d line %d\n", (int)s, __LINE__); exit(1);} } while (0)
__global__ void k_fill(float *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((x & 0xffffff) / 16777216.0f - 0.5f) * 2.0f;
    }
}
int main()
{
    hipblasLtHandle_t h; CB(hipblasLtCreate(&h));
    const size_t ws_bytes = 128u << 20; void *ws; CK(hipMalloc(&ws, ws_bytes));
    struct S { int batch, T, K, N; } shapes[] = {{72, 11440, 256, 256}, {36, 11440, 256, 512}, {36, 11440, 256, 720}};
    for (auto s : shapes) {
        const int64_t m = s.N, n = s.T, k = s.K;
        float *V, *U, *M;
        CK(hipMalloc(&V, sizeof(float) * s.batch * n * k)); CK(hipMalloc(&U, sizeof(float) * s.batch * k * m));
        CK(hipMalloc(&M, sizeof(float) * s.batch * n * m));
        k_fill<<<4096, 256>>>(V, (size_t)s.batch * n * k); k_fill<<<4096, 256>>>(U, (size_t)s.batch * k * m);
        hipblasLtMatmulDesc_t desc; CB(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t opn = HIPBLAS_OP_N;
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opn, sizeof(opn)));
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opn, sizeof(opn)));
        hipblasLtMatrixLayout_t la, lb, lc;
        CB(hipblasLtMatrixLayoutCreate(&la, HIP_R_32F, m, k, m));
        CB(hipblasLtMatrixLayoutCreate(&lb, HIP_R_32F, k, n, k));
        CB(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, m, n, m));
        int32_t bc = s.batch; int64_t sa = k * m, sb = n * k, sc = n * m;
        CB(hipblasLtMatrixLayoutSetAttribute(la, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)));
        CB(hipblasLtMatrixLayoutSetAttribute(lb, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)));
        CB(hipblasLtMatrixLayoutSetAttribute(lc, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)));
        CB(hipblasLtMatrixLayoutSetAttribute(la, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sa, sizeof(sa)));
        CB(hipblasLtMatrixLayoutSetAttribute(lb, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sb, sizeof(sb)));
        CB(hipblasLtMatrixLayoutSetAttribute(lc, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sc, sizeof(sc)));
        hipblasLtMatmulPreference_t pref; CB(hipblasLtMatmulPreferenceCreate(&pref));
        CB(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
        hipblasLtMatmulHeuristicResult_t res[16]; int nres = 0;
        CB(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, 16, res, &nres));
        printf("batch %d T %d K %d N %d: %d algos:", s.batch, s.T, s.K, s.N, nres);
        const float alpha = 1.0f, beta = 0.0f;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int a = 0; a < nres; ++a) {
            bool ok = true;
            for (int it = 0; it < 2 && ok; ++it)
                ok = hipblasLtMatmul(h, desc, &alpha, U, la, V, lb, &beta, M, lc, M, lc, &res[a].algo, ws, ws_bytes, 0) == HIPBLAS_STATUS_SUCCESS;
            if (!ok) { printf(" [a%d fail]", a); continue; }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int it = 0; it < 10; ++it)
                hipblasLtMatmul(h, desc, &alpha, U, la, V, lb, &beta, M, lc, M, lc, &res[a].algo, ws, ws_bytes, 0);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            printf(" %.3f(%.0fTF)", ms, 2.0 * s.batch * m * n * k / ms / 1e9);
        }
        printf("\n");
        CK(hipFree(V)); CK(hipFree(U)); CK(hipFree(M));
    }
    return 0;
}
