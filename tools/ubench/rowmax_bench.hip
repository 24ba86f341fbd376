// Micro-benchmark (not product code): the product's channels-last row-max kernel (csrc/decode.hip)
// on BASELINE's geometry (800x1344, batch 8, all five levels), one launch at a time, under an
// occupancy cap (unused dynamic LDS per workgroup).  Run under rocprofv3 --kernel-trace to read the
// true per-launch durations; the LDS column tells the variants apart.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         tools/ubench/rowmax_bench.hip -o tools/ubench/rowmax_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../iou-aware-single-stage-object-detector_amd/csrc/decode.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_fill(float *p, size_t n, float mu, float sd)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = mu + sd * ((float)(h & 0xffff) / 32768.f - 1.f);
    }
}

int main(int argc, char **argv)
{
    const int B = 8, iters = argc > 1 ? atoi(argv[1]) : 10;
    const int NL = argc > 2 ? atoi(argv[2]) : 5;          // pyramid levels used (from P3 down)
    ia_head_geom g;
    memset(&g, 0, sizeof(g));
    g.num_levels = NL; g.num_anchors = 9; g.num_classes = 80; g.nms_pre = 1000; g.layout = IA_LAYOUT_NHWC;
    const int H[5] = {100, 50, 25, 13, 7}, W[5] = {168, 84, 42, 21, 11}, S[5] = {8, 16, 32, 64, 128};
    for (int l = 0; l < NL; ++l) { g.H[l] = H[l]; g.W[l] = W[l]; g.stride[l] = S[l]; }
    ia::LevelTable t;
    if (ia::make_level_table(&g, t)) return 1;
    ia_level_ptrs p;
    memset(&p, 0, sizeof(p));
    size_t bytes = 0;
    for (int l = 0; l < NL; ++l) {
        const size_t rows = (size_t)B * H[l] * W[l] * 9;
        float *c, *i;
        CK(hipMalloc(&c, rows * 80 * 4)); CK(hipMalloc(&i, rows * 4));
        k_fill<<<2048, 256>>>(c, rows * 80, -4.6f, 2.f); k_fill<<<256, 256>>>(i, rows, 0.f, 1.f);
        p.cls[l] = c; p.iou[l] = i;
        bytes += rows * 81 * 4 + rows * 4;
    }
    const int N = t.anchor_off[NL];
    float *rowmax, *gm;
    CK(hipMalloc(&rowmax, (size_t)B * N * 4)); CK(hipMalloc(&gm, (size_t)B * N * 4));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int pads[] = {0, 8192, 0, 8192};
    for (int pad : pads) {
        ia::rowmax_nhwc_lds_pad = pad;
        for (int w = 0; w < 2; ++w) ia::launch_rowmax(t, p, B, IA_F32, rowmax, 0, gm);
        CK(hipDeviceSynchronize());
        float tot = 0, best = 1e9;
        for (int it = 0; it < iters; ++it) {
            CK(hipEventRecord(e0));
            ia::launch_rowmax(t, p, B, IA_F32, rowmax, 0, gm);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; best = ms < best ? ms : best;
        }
        printf("lds pad %6d B (about %2d workgroups per CU): events avg %.1f us  best %.1f us  -> %.0f GB/s\n", pad,
               (int)(160 * 1024 / (5376 + pad + 256)) > 32 ? 32 : (int)(160 * 1024 / (5376 + pad + 256)),
               tot / iters * 1e3, best * 1e3, bytes / (best * 1e-3) / 1e9);
    }
    return 0;
}
