// Micro-benchmark (not product code): does the access granularity per pixel row matter for a
// streaming read-modify-write of (P, 256) fp32 rows?  Pattern A = the accumulator layout of
// v_mfma_f32_16x16x4_f32 (a wavefront instruction touches 16 rows x 64 bytes), pattern B = 8 rows x
// 128 bytes (full cache lines), pattern C = 4 rows x 256 bytes.  Same bytes, same instruction count.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/granule_bench.hip -o tools/ubench/granule_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(4))) float f4;
template <int ROWS>   // rows per wavefront instruction: 16, 8 or 4; 16 instructions per 16-row tile
__global__ void __launch_bounds__(512) k(const float *r, float *y, int tiles)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 8 + (threadIdx.x >> 6), nw = gridDim.x * 8;
    constexpr int LPR = 64 / ROWS;                 // lanes per row
    const int row_in = lane / LPR, piece = lane % LPR;
    for (int t = wave; t < tiles; t += nw) {
        f4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // instruction i: rows (i % (16 / ROWS)) * ROWS + row_in, columns ((i / (16 / ROWS)) * LPR + piece) * 4
            const int row = (i % (16 / ROWS)) * ROWS + row_in, col = ((i / (16 / ROWS)) * LPR + piece) * 4;
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(r + ((size_t)t * 16 + row) * 256 + col));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i % (16 / ROWS)) * ROWS + row_in, col = ((i / (16 / ROWS)) * LPR + piece) * 4;
            f4 w = v[i]; w.x = w.x > 0.f ? w.x : 0.f; w.y += 1.0f;
            *reinterpret_cast<f4 *>(y + ((size_t)t * 16 + row) * 256 + col) = w;
        }
    }
}
int main()
{
    const size_t P = 8ull * 200 * 336;
    float *r, *y; CK(hipMalloc(&r, P * 256 * 4)); CK(hipMalloc(&y, P * 256 * 4));
    CK(hipMemset(r, 0, P * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int tiles = (int)(P / 16);
    auto run = [&](int rows, const char *name) {
        for (int rep = 0; rep < 23; ++rep) {
            if (rep == 3) CK(hipEventRecord(e0));
            if (rows == 16) hipLaunchKernelGGL(k<16>, dim3(512), dim3(512), 0, 0, r, y, tiles);
            else if (rows == 8) hipLaunchKernelGGL(k<8>, dim3(512), dim3(512), 0, 0, r, y, tiles);
            else hipLaunchKernelGGL(k<4>, dim3(512), dim3(512), 0, 0, r, y, tiles);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("%-28s %.1f us  %.2f TB/s\n", name, ms * 1e3, 2.0 * P * 256 * 4 / ms / 1e9);
    };
    run(16, "16 rows x 64 B per instruction"); run(8, "8 rows x 128 B"); run(4, "4 rows x 256 B");
    run(16, "16 rows x 64 B (again)");
    return 0;
}
