// nms_bench.hip -- nms_kernel on synthetic candidate sets (never part of libzly.so): how long does one frame take as a function of the number
// of candidates and of how they spread over classes?  (The synthetic YOLOv8-s at 640 x 640: ~600-850 candidates, 90 % in one class.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Izero-latency-yolo_amd/csrc zero-latency-yolo_amd/tools/nms_bench.hip -o zero-latency-yolo_amd/_build/nms_bench
#include "../csrc/kernels_post.hip"
#include <stdio.h>
#include <vector>
#include <random>
using namespace zly;

#ifdef ZLY_NMS_DIAG
static unsigned long long* g_dbg = nullptr;
#endif
static void run(const char* name, int frames, int n, int big_class_share_pct, int nc, float spread)
{
    const int N = 8400, cap = 1024;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<Cand> h((size_t)frames * N);
    std::vector<int> cnt(frames, n);
    for (int f = 0; f < frames; ++f)
        for (int i = 0; i < n; ++i) {
            Cand c;
            c.x = U(rng) * spread; c.y = U(rng) * spread; c.w = 0.05f + 0.1f * U(rng); c.h = 0.05f + 0.1f * U(rng);
            c.conf = 0.5f + 0.5f * U(rng);
            c.cls = (int)(U(rng) * 100) < big_class_share_pct ? 0 : 1 + (int)(U(rng) * (nc - 1)) % (nc - 1);
            c.anchor = i; c.pad_ = 0;
            h[(size_t)f * N + i] = c;
        }
    Cand *d, *scratch; int* dc; void* slabs;
    const size_t slab = sizeof(zly_slab_header) + (size_t)cap * sizeof(zly_det);
    hipMalloc((void**)&d, h.size() * sizeof(Cand)); hipMalloc((void**)&scratch, h.size() * sizeof(Cand)); hipMalloc((void**)&dc, frames * 4); hipMalloc(&slabs, slab * frames);
    hipMemcpy(d, h.data(), h.size() * sizeof(Cand), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
        hipMemcpy(dc, cnt.data(), frames * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0, 0);
        nms_init(); launch_nms(d, dc, N, frames, 0.45f, nc, scratch, slabs, cap, 0, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    zly_slab_header hd; hipMemcpy(&hd, slabs, sizeof hd, hipMemcpyDeviceToHost);
#ifdef ZLY_NMS_DIAG
    {
        unsigned long long h7[8] = {0}; hipMemcpy(h7, g_dbg, sizeof h7, hipMemcpyDeviceToHost);
        if (h7[6] > h7[0]) printf("      general path, frame 0, cycles: key staging %llu | rank sort %llu | class bounds %llu | crowded classes %llu | other classes %llu | compaction %llu\n",
                                  h7[1] - h7[0], h7[2] - h7[1], h7[3] - h7[2], h7[4] - h7[3], h7[5] - h7[4], h7[6] - h7[5]);
        hipMemset(g_dbg, 0, 64);
    }
#endif
    printf("%-44s %2d frames x %4d candidates, %3d %% in one class: %8.1f us   (frame 0 kept %d)\n", name, frames, n, big_class_share_pct, best * 1e3, hd.n_kept);
    hipFree(d); hipFree(scratch); hipFree(dc); hipFree(slabs);
}

int main()
{
#ifdef ZLY_NMS_DIAG
    hipMalloc((void**)&g_dbg, 64); hipMemset(g_dbg, 0, 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_nms_diag), &g_dbg, sizeof g_dbg);
#endif
    run("one-wave path", 32, 100, 20, 80, 1.0f);
    run("class segments in registers (all <= 64)", 32, 600, 0, 80, 1.0f);
    run("crowded class, sparse boxes", 32, 600, 90, 80, 1.0f);
    run("crowded class, dense boxes", 32, 600, 90, 80, 0.3f);
    run("crowded class, 850", 32, 850, 90, 80, 0.5f);
    run("crowded class, 850, one frame", 1, 850, 90, 80, 0.5f);
    run("1100 candidates (global-memory path)", 32, 1100, 90, 80, 0.5f);
    run("1100 candidates, spread over classes", 32, 1100, 0, 80, 0.5f);
    return 0;
}
