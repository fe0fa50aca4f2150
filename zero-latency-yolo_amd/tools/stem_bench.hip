// stem_bench.hip -- DIAGNOSTIC build of stem_model1_kernel (preprocess + model.0 + model.1) with s_memtime stamps around its phases
// (never part of libzly.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Izero-latency-yolo_amd/csrc -DZLY_STEM_DIAG=1 zero-latency-yolo_amd/tools/stem_bench.hip \
//         -o zero-latency-yolo_amd/_build/stem_bench && ./zero-latency-yolo_amd/_build/stem_bench
// YOLOv8n at 416 x 416, batch 64 (and 1): per variant (0 = round 3's staging / tap order, 1 = conflict-free, 2 = persistent workgroups with the next tile's input in flight) and waves per workgroup the launch time and
// the per-wave cycle sums of: prologue | patch staging (loads, convert, LDS stores) | barrier | stem conv -> LDS map | barrier | model.1 -> HBM.
// Random weights: the timing does not depend on the values (the parity tests check them).  Read SHARES, not the stamped build's length.
#include "../csrc/kernels_stem.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
using namespace zly;

static void* dalloc_bf16(size_t elems, unsigned seed)
{
    std::vector<uint16_t> h(elems);
    for (size_t i = 0; i < elems; ++i) h[i] = (uint16_t)(0x3800 + ((((unsigned)i * 2654435761u + seed) >> 20) & 0x3ff)) ^ (uint16_t)((i & 1) << 15);
    void* d; (void)hipMalloc(&d, elems * 2); (void)hipMemcpy(d, h.data(), elems * 2, hipMemcpyHostToDevice);
    return d;
}

static void run(int n, int var, int nw, int th, int tw, int pgrid = 0)
{
    const int W = 416, H = 416;
    std::vector<uint8_t> hf((size_t)n * W * H * 3);
    for (size_t i = 0; i < hf.size(); ++i) hf[i] = (uint8_t)((i * 2654435761u) >> 24);
    uint8_t* dsrc; (void)hipMalloc((void**)&dsrc, hf.size()); (void)hipMemcpy(dsrc, hf.data(), hf.size(), hipMemcpyHostToDevice);
    std::vector<FrameDesc> hd((size_t)n);
    for (int i = 0; i < n; ++i) { hd[(size_t)i].src_off = (unsigned long long)i * W * H * 3; hd[(size_t)i].w = W; hd[(size_t)i].h = H; }
    FrameDesc* ddesc; (void)hipMalloc((void**)&ddesc, hd.size() * sizeof(FrameDesc)); (void)hipMemcpy(ddesc, hd.data(), hd.size() * sizeof(FrameDesc), hipMemcpyHostToDevice);
    float* bias; (void)hipMalloc((void**)&bias, 64 * 4); (void)hipMemset(bias, 0, 64 * 4);
    Stem1Args a; memset(&a, 0, sizeof a);
    a.st.src = dsrc; a.st.desc = ddesc; a.st.wgt = dalloc_bf16(1024, 1); a.st.bias = bias;
    void* out0; (void)hipMalloc(&out0, (size_t)n * 208 * 208 * 16 * 2); a.st.out = out0; a.st.out_cs = 16;
    a.st.tw = W; a.st.th = H; a.st.Ho = 208; a.st.Wo = 208; a.st.Cout = 16;
    a.w1 = dalloc_bf16(2 * 9 * 256, 2); a.b1 = bias;
    void* out1; (void)hipMalloc(&out1, (size_t)n * 104 * 104 * 32 * 2); a.out1 = out1; a.out1_cs = 32;
    a.H1 = 104; a.W1 = 104; a.TH = th; a.TW = tw; a.tiles_x = (104 + tw - 1) / tw; a.tiles_y = (104 + th - 1) / th;
    a.wgt0p = a.st.wgt; a.nw = nw; a.var = var; a.pgrid = pgrid;
    const size_t nwaves = (size_t)a.tiles_x * a.tiles_y * n * nw;
    unsigned long long* ddbg; (void)hipMalloc((void**)&ddbg, nwaves * 64); (void)hipMemset(ddbg, 0, nwaves * 64);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stem_diag), &ddbg, sizeof ddbg);
    (void)stem1_init();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 30; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (launch_stem_model1(a, n, 0) != hipSuccess) { printf("launch failed (var %d nw %d tile %dx%d)\n", var, nw, th, tw); return; }
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 10 && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(nwaves * 8);
    (void)hipMemcpy(h.data(), ddbg, nwaves * 64, hipMemcpyDeviceToHost);
    double s[8] = {0}; for (size_t w = 0; w < nwaves; ++w) for (int k = 0; k < 7; ++k) s[k] += (double)h[w * 8 + k];
    printf("batch %2d var %d grid %4d, %2d waves, tile %dx%d (%d tiles): %6.1f us best of 20 | cycles per wave (mean): prologue %.0f | staging %.0f | barrier %.0f | stem conv %.0f | barrier %.0f | model.1 %.0f | wave total %.0f\n",
           n, var, pgrid, nw, th, tw, a.tiles_x * a.tiles_y * n, best * 1e3, s[0] / nwaves, s[1] / nwaves, s[2] / nwaves, s[3] / nwaves, s[4] / nwaves, s[5] / nwaves, s[6] / nwaves);
    (void)hipFree(dsrc); (void)hipFree(ddesc); (void)hipFree(bias); (void)hipFree(out0); (void)hipFree(out1); (void)hipFree(ddbg);
    (void)hipFree(const_cast<void*>(a.st.wgt)); (void)hipFree(const_cast<void*>(a.w1));
}

int main()
{
    // cycles are sums over a wave's tiles divided by ALL tiles x waves: per tile and wave in every variant (var 2: a workgroup walks several tiles)
    for (int var = 0; var <= 2; ++var) {
        run(64, var, 8, 8, 26);
        run(64, var, 16, 8, 26);
        run(64, var, 8, 4, 26);
        run(1, var, 8, 8, 26);
    }
    for (int g : {256, 416, 476, 512, 555, 666, 768, 832}) run(64, 2, 8, 8, 26, g);
    return 0;
}
