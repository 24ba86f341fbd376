// Micro-benchmark (not product code): the bottleneck's last 1x1 convolution as ONE hipBLASLt GEMM
// with the folded-BN bias, the residual (beta * C) and the ReLU in its epilogue:
//   D (pixels x Cout, row-major) = relu(A (pixels x Cin) . W (Cin x Cout) + residual + bias)
// hipcc tools/ubench/lt_conv3.cpp -lhipblaslt -o tools/ubench/lt_conv3
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define CB(x) do { hipblasStatus_t s = (x); if (s != HIPBLAS_STATUS_SUCCESS) { printf("hipblaslt error %d line %d\n", (int)s, __LINE__); exit(1);} } while (0)

int main()
{
    hipblasLtHandle_t h; CB(hipblasLtCreate(&h));
    const size_t ws_bytes = 64u << 20; void *ws; CK(hipMalloc(&ws, ws_bytes));
    struct S { int pix, cin, cout; } shapes[] = {{8 * 200 * 336, 64, 256}, {8 * 100 * 168, 128, 512},
                                                 {8 * 50 * 84, 256, 1024}, {8 * 25 * 42, 512, 2048},
                                                 {8 * 200 * 336, 256, 64}, {8 * 50 * 84, 1024, 256}};
    for (auto s : shapes) {
        // column-major view: D^T (Cout x pix) = W^T (Cout x Cin) . A^T (Cin x pix)
        const int64_t m = s.cout, n = s.pix, k = s.cin;
        float *A, *W, *R, *D, *bias;
        CK(hipMalloc(&A, sizeof(float) * n * k)); CK(hipMalloc(&W, sizeof(float) * k * m));
        CK(hipMalloc(&R, sizeof(float) * n * m)); CK(hipMalloc(&D, sizeof(float) * n * m));
        CK(hipMalloc(&bias, sizeof(float) * m));
        CK(hipMemset(A, 0, sizeof(float) * n * k)); CK(hipMemset(W, 0, sizeof(float) * k * m));
        CK(hipMemset(R, 0, sizeof(float) * n * m)); CK(hipMemset(bias, 0, sizeof(float) * m));
        hipblasLtMatmulDesc_t desc; CB(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t opn = HIPBLAS_OP_N;
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opn, sizeof(opn)));
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opn, sizeof(opn)));
        hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_RELU_BIAS;
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
        // W stored (Cin x Cout) row-major == column-major (Cout x Cin) with ld = Cout
        hipblasLtMatrixLayout_t la, lb, lc, ld;
        CB(hipblasLtMatrixLayoutCreate(&la, HIP_R_32F, m, k, m));
        CB(hipblasLtMatrixLayoutCreate(&lb, HIP_R_32F, k, n, k));
        CB(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, m, n, m));
        CB(hipblasLtMatrixLayoutCreate(&ld, HIP_R_32F, m, n, m));
        hipblasLtMatmulPreference_t pref; CB(hipblasLtMatmulPreferenceCreate(&pref));
        CB(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
        hipblasLtMatmulHeuristicResult_t res[8]; int nres = 0;
        CB(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, ld, pref, 8, res, &nres));
        printf("pix %7d  %4d -> %4d : %d algos;", s.pix, s.cin, s.cout, nres);
        const float alpha = 1.0f, beta = 1.0f;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int a = 0; a < nres && a < 4; ++a) {
            for (int it = 0; it < 3; ++it)
                CB(hipblasLtMatmul(h, desc, &alpha, W, la, A, lb, &beta, R, lc, D, ld, &res[a].algo, ws, ws_bytes, 0));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            const int iters = 20;
            for (int it = 0; it < iters; ++it)
                CB(hipblasLtMatmul(h, desc, &alpha, W, la, A, lb, &beta, R, lc, D, ld, &res[a].algo, ws, ws_bytes, 0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
            printf("  algo%d %.3f ms (%.0f TF)", a, ms, 2.0 * m * n * k / ms / 1e9);
        }
        printf("\n");
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(R)); CK(hipFree(D)); CK(hipFree(bias));
    }
    return 0;
}
