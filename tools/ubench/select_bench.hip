// Micro-benchmark (not product code): the top-k kernels of csrc/select.hip on a synthetic
// row-max array of BASELINE's geometry (800x1344, batch 8, channels-last order), with in-kernel
// phase timestamps (IA_SEL_PROFILE) for the first image's segments.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DIA_SEL_PROFILE \
//         tools/ubench/select_bench.hip -o tools/ubench/select_bench && tools/ubench/select_bench [A|D] [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../../iou-aware-single-stage-object-detector_amd/csrc/decode.hip"
#include "../../iou-aware-single-stage-object-detector_amd/csrc/select.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char **argv)
{
    const char kind = argc > 1 ? argv[1][0] : 'D';
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const int B = 8;
    ia_head_geom g;
    memset(&g, 0, sizeof(g));
    g.num_levels = 5; g.num_anchors = 9; g.num_classes = 80; g.nms_pre = 1000; g.layout = IA_LAYOUT_NHWC;
    const int H[5] = {100, 50, 25, 13, 7}, W[5] = {168, 84, 42, 21, 11}, S[5] = {8, 16, 32, 64, 128};
    for (int l = 0; l < 5; ++l) { g.H[l] = H[l]; g.W[l] = W[l]; g.stride[l] = S[l]; }
    ia::LevelTable t;
    if (ia::make_level_table(&g, t)) { printf("bad geometry\n"); return 1; }
    const int N = t.anchor_off[5], R = t.cand_off[5];
    std::vector<float> h((size_t)B * N);
    srand(1);
    for (auto &v : h) {
        const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
        const double z = sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2);
        if (kind == 'D') v = (float)(0.0707 * (1.0 + 3e-4 * z));          // random-init-like near-ties
        else { const double x = -6 + 2 * z + 4.6; v = (float)sqrt(1 / (1 + exp(-x)) * 0.5); }   // spread
    }
    float *rowmax; int32_t *cand; void *ws;
    const size_t wsb = ia::select_workspace_bytes(t, B);
    CK(hipMalloc(&rowmax, h.size() * 4)); CK(hipMalloc(&cand, (size_t)B * R * 4)); CK(hipMalloc(&ws, wsb));
    CK(hipMemcpy(rowmax, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("kind %c  N %d  R %d  workspace %.1f MB\n", kind, N, R, wsb / 1e6);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (ia::launch_select(t, rowmax, B, cand, ws, 0, false)) { printf("launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) ia::launch_select(t, rowmax, B, cand, ws, 0, false);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("launch_select (groupmax + filter + final): %.1f us per call\n", ms * 1e3 / iters);
    // check against a host sort (image 0)
    std::vector<int32_t> hc((size_t)B * R);
    CK(hipMemcpy(hc.data(), cand, hc.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int b = 0; b < B; ++b)
        for (int l = 0; l < 5; ++l) {
            const int n = t.anchor_off[l + 1] - t.anchor_off[l], k = t.cand_off[l + 1] - t.cand_off[l];
            std::vector<int> idx(n);
            for (int i = 0; i < n; ++i) idx[i] = i;
            const float *src = h.data() + (size_t)b * N + t.anchor_off[l];
            if (k < n) std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return src[x] > src[y]; });
            for (int i = 0; i < k; ++i) bad += hc[(size_t)b * R + t.cand_off[l] + i] != idx[i];
        }
    printf("mismatches vs host stable sort: %d\n", bad);
    unsigned long long prof[2][IA_MAX_LEVELS][24];
    CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(ia::g_sel_prof), sizeof(prof)));
    {   // chunk counts sit at the start of the workspace: (B, 47) for this geometry
        std::vector<uint32_t> cnt(47);
        CK(hipMemcpy(cnt.data(), ws, sizeof(uint32_t) * 47, hipMemcpyDeviceToHost));
        uint32_t p3 = 0, p4 = 0;
        for (int c = 0; c < 37; ++c) p3 += cnt[c];
        for (int c = 37; c < 47; ++c) p4 += cnt[c];
        printf("candidates of image 0: P3 %u  P4 %u\n", p3, p4);
    }
    const char *names[2] = {"k_sel_filter (first chunk of the level)", "k_sel_final"};
    for (int kk = 0; kk < 2; ++kk) {
        printf("%s, image 0 -- phase deltas in us (100 MHz wall clock):\n", names[kk]);
        for (int l = 0; l < 5; ++l) {
            printf("  level %d:", l);
            unsigned long long last = prof[kk][l][0];
            for (int i = 1; i < 24; ++i)
                if (prof[kk][l][i]) { printf("  [%d] +%.2f", i, (double)(prof[kk][l][i] - last) * 0.01); last = prof[kk][l][i]; }
            printf("   total %.2f\n", (double)(last - prof[kk][l][0]) * 0.01);
        }
    }
    {
        static unsigned long long blk[2][16][64][2];
        CK(hipMemcpyFromSymbol(blk, HIP_SYMBOL(ia::g_sel_blk), sizeof(blk)));
        unsigned long long t0 = ~0ull, t1 = 0; double sum = 0, mx = 0; int nb = 0;
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < 47; ++c) {
                const unsigned long long s = blk[0][b][c][0], e = blk[0][b][c][1];
                t0 = s < t0 ? s : t0; t1 = e > t1 ? e : t1;
                sum += (e - s) * 0.01; mx = (e - s) * 0.01 > mx ? (e - s) * 0.01 : mx; ++nb;
            }
        printf("k_sel_filter blocks: first start -> last end %.2f us; block time avg %.2f max %.2f us; last start - first start %.2f us\n",
               (t1 - t0) * 0.01, sum / nb, mx, 0.0);
        unsigned long long smax = 0;
        for (int b = 0; b < B; ++b) for (int c = 0; c < 47; ++c) smax = blk[0][b][c][0] > smax ? blk[0][b][c][0] : smax;
        printf("  start skew (last start - first start) %.2f us\n", (smax - t0) * 0.01);
    }
    return bad != 0;
}
