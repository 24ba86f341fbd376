// Micro-benchmark (not product code): the channels-last gather / decode kernel of csrc/decode.hip on
// BASELINE's geometry (batch 8, 4693 candidates per image), with in-kernel phase timestamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         -DIA_GATHER_PROFILE tools/ubench/gather_bench.hip -o tools/ubench/gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../iou-aware-single-stage-object-detector_amd/csrc/decode.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_touch(int32_t *p, size_t n)       // rewrites the candidate list like the top-k kernel would
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] + 0;
}

__global__ void k_fill(float *p, size_t n, float mu, float sd)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = mu + sd * ((float)(h & 0xffff) / 32768.f - 1.f);
    }
}

int main(int argc, char **argv)
{
    const int B = 8, iters = argc > 1 ? atoi(argv[1]) : 20;
    ia_head_geom g;
    memset(&g, 0, sizeof(g));
    g.num_levels = 5; g.num_anchors = 9; g.num_classes = 80; g.nms_pre = 1000; g.layout = IA_LAYOUT_NHWC;
    const int H[5] = {100, 50, 25, 13, 7}, W[5] = {168, 84, 42, 21, 11}, S[5] = {8, 16, 32, 64, 128};
    for (int l = 0; l < 5; ++l) { g.H[l] = H[l]; g.W[l] = W[l]; g.stride[l] = S[l]; }
    for (int l = 0; l < 5; ++l) for (int a = 0; a < 9; ++a) for (int k = 0; k < 4; ++k)
        g.base_anchors[l][a][k] = (k < 2 ? -1.f : 1.f) * S[l] * (1 + a);
    for (int k = 0; k < 4; ++k) { g.means[k] = 0; g.stds[k] = 1; }
    ia::LevelTable t;
    if (ia::make_level_table(&g, t)) return 1;
    ia::BaseAnchors ba;
    memcpy(ba.v, g.base_anchors, sizeof(ba.v));
    ia_level_ptrs p;
    memset(&p, 0, sizeof(p));
    for (int l = 0; l < 5; ++l) {
        const size_t rows = (size_t)B * H[l] * W[l] * 9;
        float *c, *i, *r;
        CK(hipMalloc(&c, rows * 80 * 4)); CK(hipMalloc(&i, rows * 4)); CK(hipMalloc(&r, rows * 16));
        k_fill<<<2048, 256>>>(c, rows * 80, -4.6f, 2.f); k_fill<<<256, 256>>>(i, rows, 0.f, 1.f);
        k_fill<<<256, 256>>>(r, rows * 4, 0.f, .5f);
        p.cls[l] = c; p.iou[l] = i; p.reg[l] = r;
    }
    const int R = t.cand_off[5], Rs = (R + 63) / 64 * 64;
    std::vector<int32_t> hc((size_t)B * R);
    srand(3);
    for (int b = 0; b < B; ++b)
        for (int l = 0; l < 5; ++l) {
            const int n = t.anchor_off[l + 1] - t.anchor_off[l], k = t.cand_off[l + 1] - t.cand_off[l];
            for (int i = 0; i < k; ++i) hc[(size_t)b * R + t.cand_off[l] + i] = k == n ? i : (int)(((long long)rand() * 7919 + i) % n);
        }
    int32_t *cand; float *boxes, *scores, *best, *hw, *sf;
    CK(hipMalloc(&cand, hc.size() * 4)); CK(hipMemcpy(cand, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&boxes, (size_t)B * R * 16)); CK(hipMalloc(&scores, (size_t)B * 80 * Rs * 4)); CK(hipMalloc(&best, (size_t)B * R * 4));
    std::vector<float> hhw(2 * B), hsf(4 * B, 1.f);
    for (int b = 0; b < B; ++b) { hhw[2 * b] = 800; hhw[2 * b + 1] = 1333; }
    CK(hipMalloc(&hw, 8 * B)); CK(hipMalloc(&sf, 16 * B));
    CK(hipMemcpy(hw, hhw.data(), 8 * B, hipMemcpyHostToDevice)); CK(hipMemcpy(sf, hsf.data(), 16 * B, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    for (int lds : {0, 10560, 21120}) {
        int nb = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ia::k_gather_nhwc<float, 2, 5>, 320, lds));
        printf("occupancy API: 320 threads, %d B dynamic LDS -> %d blocks per CU\n", lds, nb);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&] { return ia::launch_gather(t, ba, g.means, g.stds, p, B, IA_F32, cand, hw, sf, 1, boxes, scores, best, Rs, 0); };
    for (int i = 0; i < 3; ++i) if (run()) { printf("launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    float *flush; const size_t fl = (size_t)1 << 28;      // 1 GiB: evicts L2 and the Infinity Cache
    CK(hipMalloc(&flush, fl * 4));
    const bool cold = argc > 2;
    float tot = 0;
    for (int i = 0; i < iters; ++i) {
        if (cold) { k_fill<<<4096, 256>>>(flush, fl, 0.f, 1.f); k_touch<<<40, 1024>>>(cand, (size_t)B * R); }
        CK(hipEventRecord(e0)); run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
    }
    printf("launch_gather (%s caches): %.1f us per call (events, one launch at a time)\n", cold ? "cold" : "warm", tot / iters * 1e3);
#ifdef IA_GATHER_PROFILE
    unsigned long long prof[4][16];
    CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(ia::g_gather_prof), sizeof(prof)));
    for (int k = 0; k < 3; ++k) {
        printf("block (%d, 3):", k * 49);
        for (int i = 1; i < 6; ++i) printf("  [%d] +%.2f", i, (double)(prof[k][i] - prof[k][i - 1]) * 0.01);
        printf("   total %.2f us\n", (double)(prof[k][5] - prof[k][0]) * 0.01);
    }
    static unsigned long long blk[8][160][4];
    CK(hipMemcpyFromSymbol(blk, HIP_SYMBOL(ia::g_gather_blk), sizeof(blk)));
    unsigned long long t0 = ~0ull, t1 = 0, smax = 0; double sum = 0, mx = 0;
    const int nbx = (R + 31) / 32;
    for (int b = 0; b < B; ++b) for (int x = 0; x < nbx; ++x) {
        const unsigned long long s = blk[b][x][0], e = blk[b][x][1] > blk[b][x][2] ? blk[b][x][1] : blk[b][x][2];
        t0 = s < t0 ? s : t0; t1 = e > t1 ? e : t1; smax = s > smax ? s : smax;
        sum += (e - s) * 0.01; mx = (e - s) * 0.01 > mx ? (e - s) * 0.01 : mx;
    }
    printf("blocks: first start -> last end %.2f us; start skew %.2f us; block time avg %.2f max %.2f us\n",
           (t1 - t0) * 0.01, (smax - t0) * 0.01, sum / (B * nbx), mx);
    // histogram of block start times
    int hist[16] = {0};
    for (int b = 0; b < B; ++b) for (int x = 0; x < nbx; ++x) { int h = (int)((blk[b][x][0] - t0) * 0.01); hist[h > 15 ? 15 : h]++; }
    printf("block starts per us:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\n");
#endif
    return 0;
}
