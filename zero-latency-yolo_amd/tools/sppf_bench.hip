// sppf_bench.hip -- DIAGNOSTIC build of sppf_fused_kernel with s_memtime stamps around its phases (never part of libzly.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Izero-latency-yolo_amd/csrc -DZLY_SPPF_DIAG=1 zero-latency-yolo_amd/tools/sppf_bench.hip \
//         -o zero-latency-yolo_amd/_build/sppf_bench && ./zero-latency-yolo_amd/_build/sppf_bench
// YOLOv8n's SPPF at 13 x 13 (256 -> 128 -> 256), batch 64 and 1, split 4 and 2.  Per-wave cycle sums: x DMA + cv1 weights | cv1 | y -> sortable |
// cv2 phases (4) | row passes (3) | barrier | column passes (3) | barrier.  Random weights; read SHARES, not the stamped build's length.
#include "../csrc/kernels_sppf.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
namespace zly { int num_cus() { return 256; } }
using namespace zly;

static void* dalloc_bf16(size_t elems, unsigned seed)
{
    std::vector<uint16_t> h(elems);
    for (size_t i = 0; i < elems; ++i) h[i] = (uint16_t)(0x3800 + ((((unsigned)i * 2654435761u + seed) >> 20) & 0x3ff)) ^ (uint16_t)((i & 1) << 15);
    void* d; (void)hipMalloc(&d, elems * 2); (void)hipMemcpy(d, h.data(), elems * 2, hipMemcpyHostToDevice);
    return d;
}

static void run(int n, int split)
{
    SppfArgs a; memset(&a, 0, sizeof a);
    const int HW = 169;
    a.x = dalloc_bf16((size_t)n * HW * 256, 1); a.x_cs = 256; a.Cin = 256;
    a.w1 = dalloc_bf16(128 * 256, 2); a.w2 = dalloc_bf16(256 * 512, 3);
    float* bias; (void)hipMalloc((void**)&bias, 512 * 4); (void)hipMemset(bias, 0, 512 * 4);
    a.b1 = bias; a.b2 = bias;
    void* out; (void)hipMalloc(&out, (size_t)n * HW * 384 * 2); a.out = out; a.out_cs = 384; a.out_co = 128; a.Cout = 256;
    void* cat; (void)hipMalloc(&cat, (size_t)n * HW * 512 * 2); a.cat = cat; a.cat_cs = 512;
    a.H = 13; a.W = 13; a.n = n; a.c = 128; a.split = split; a.dump = 0;
    const size_t nwaves = (size_t)n * split * 16;
    unsigned long long* ddbg; (void)hipMalloc((void**)&ddbg, nwaves * 128); (void)hipMemset(ddbg, 0, nwaves * 128);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sppf_diag), &ddbg, sizeof ddbg);
    (void)sppf_init();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 30; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (launch_sppf_fused(a, 0) != hipSuccess) { printf("launch failed\n"); return; }
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 10 && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(nwaves * 16);
    (void)hipMemcpy(h.data(), ddbg, nwaves * 128, hipMemcpyDeviceToHost);
    double s[9] = {0}; for (size_t w = 0; w < nwaves; ++w) for (int k = 0; k < 9; ++k) s[k] += (double)h[w * 16 + k];
    printf("batch %2d split %d (%d workgroups of 16 waves): %6.1f us best of 20 | cycles per wave (mean): dma (x, w1) %.0f | cv1 %.0f | barrier + y -> T0 + barrier %.0f | cv2 x4 %.0f | row pass x3 %.0f | barrier x3 %.0f | col pass x3 %.0f | barrier x3 %.0f | total %.0f\n",
           n, split, n * split, best * 1e3, s[0] / nwaves, s[1] / nwaves, s[2] / nwaves, s[3] / nwaves, s[5] / nwaves, s[4] / nwaves, s[6] / nwaves, s[7] / nwaves, s[8] / nwaves);
}

int main()
{
    run(64, 4); run(64, 2); run(1, 4); run(1, 2); run(16, 4);
    return 0;
}
