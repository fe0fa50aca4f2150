// c2f_bench.hip -- DIAGNOSTIC build of c2f_kernel (16 / 32 channels) with s_memtime stamps around its phases (never part of libzly.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Izero-latency-yolo_amd/csrc -DZLY_C2F_DIAG=1 zero-latency-yolo_amd/tools/c2f_bench.hip \
//         -o zero-latency-yolo_amd/_build/c2f_bench && ./zero-latency-yolo_amd/_build/c2f_bench
// Blocks of YOLOv8n at 416 x 416, batch 64: model.2 (C = 16, whole block), model.4 front / back (C = 32), model.15 (C = 32, whole, dual source).
// Per launch: time, and per wave and tile the cycle sums of: prologue | tile barrier | cv1 loop | barrier | conv A loop | barrier | conv B loop | barrier | cv2 loop.
#include "../csrc/kernels_pair.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
namespace zly {
int num_cus() { return 256; }
bool c2f64_plan(int, int, int, int, int, int, int, C2fPlan*) { return false; }
hipError_t c2f64_init() { return hipSuccess; }
hipError_t launch_c2f64(int, const C2fArgs&, const C2fPlan&, hipStream_t) { return hipErrorInvalidValue; }
}
using namespace zly;

static void* dalloc_rand(size_t elems, unsigned seed, unsigned short base, unsigned mask)
{
    std::vector<uint16_t> h(elems);
    for (size_t i = 0; i < elems; ++i) h[i] = (uint16_t)(base + (((unsigned)i * 2654435761u + seed) >> 20 & mask)) ^ (uint16_t)((i & 1) << 15);
    void* d; (void)hipMalloc(&d, elems * 2); (void)hipMemcpy(d, h.data(), elems * 2, hipMemcpyHostToDevice);
    return d;
}

static void run(const char* name, int c, int mode, int n, int H, int W, int cin, bool dual, int nmaps, int cout2)
{
    const int nk1 = cin / 32, nk2 = nmaps;
    C2fPlan pl{};
    if (!c2f_plan(c, mode, nk1, nk2, cout2, n, H, W, &pl)) { printf("%s: no plan\n", name); return; }
    C2fArgs a; memset(&a, 0, sizeof a);
    const size_t px = (size_t)n * H * W;
    if (dual) { a.x = dalloc_rand(px / 4 * 128, 1, 0x3c00, 0x1ff); a.x_cs = 128; a.x2 = dalloc_rand(px * 64, 2, 0x3c00, 0x1ff); a.x2_cs = 64; a.split_c = 128; }
    else { a.x = dalloc_rand(px * cin, 1, 0x3c00, 0x1ff); a.x_cs = cin; }
    a.w1 = dalloc_rand((size_t)(2 * c / 16) * (nk1 ? nk1 : 1) * 512, 3, 0x3400, 0x3ff); a.nk1 = nk1;
    a.wA = dalloc_rand((size_t)9 * (c / 16) * 512, 4, 0x3000, 0x3ff); a.wB = dalloc_rand((size_t)9 * (c / 16) * 512, 5, 0x3000, 0x3ff);
    a.w2 = dalloc_rand((size_t)(cout2 / 16) * nk2 * 512, 6, 0x3000, 0x3ff); a.nk2 = nk2; a.Cout2 = cout2;
    float* bias; (void)hipMalloc((void**)&bias, 512 * 4); (void)hipMemset(bias, 0, 512 * 4);
    a.b1 = bias; a.bA = bias; a.bB = bias; a.b2 = bias;
    a.cat = dalloc_rand(px * c * nmaps, 7, 0x3c00, 0x1ff); a.cat_cs = c * nmaps;
    a.pair_in_co = mode == 2 ? 2 * c : c; a.pair_out_co = a.pair_in_co + c; a.res = 1;
    void* out; (void)hipMalloc(&out, px * cout2 * 2); a.out = out; a.out_cs = cout2;
    a.H = H; a.W = W; a.n = n; a.TH = pl.th; a.TW = pl.tw; a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.total_tiles = pl.total_tiles;
    const int nw = c == 16 ? 16 : (mode == 1 ? 8 : 16);
    const size_t nwaves = (size_t)pl.grid * nw;
    unsigned long long* ddbg; (void)hipMalloc((void**)&ddbg, nwaves * 128); (void)hipMemset(ddbg, 0, nwaves * 128);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_c2f_diag), &ddbg, sizeof ddbg);
    (void)c2f_init();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 30; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (launch_c2f(c, mode, a, pl, 0) != hipSuccess) { printf("%s: launch failed\n", name); return; }
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 10 && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(nwaves * 16);
    (void)hipMemcpy(h.data(), ddbg, nwaves * 128, hipMemcpyDeviceToHost);
    double s[10] = {0}; for (size_t w = 0; w < nwaves; ++w) for (int k = 0; k < 10; ++k) s[k] += (double)h[w * 16 + k];
    const double tpw = (double)pl.total_tiles / pl.grid;
    printf("%-30s tile %dx%d, %d tiles on %d workgroups of %d waves, lds %d KB: %.1f us best of 20\n   cycles per wave (mean): prologue %.0f | then per tile: barrier %.0f | cv1 %.0f | barrier %.0f | conv A %.0f | barrier %.0f | conv B %.0f | barrier %.0f | cv2 %.0f | wave total %.0f (%.1f tiles per workgroup)\n",
           name, pl.th, pl.tw, pl.total_tiles, pl.grid, nw, pl.lds_bytes / 1024, best * 1e3,
           s[0] / nwaves, s[1] / nwaves / tpw, s[2] / nwaves / tpw, s[3] / nwaves / tpw, s[4] / nwaves / tpw, s[5] / nwaves / tpw, s[6] / nwaves / tpw, s[7] / nwaves / tpw, s[8] / nwaves / tpw, s[9] / nwaves, tpw);
}

int main()
{
    run("model.2 (C=16, 32->32) x64", 16, 3, 64, 104, 104, 32, false, 3, 32);
    run("model.4 front (C=32) x64", 32, 1, 64, 52, 52, 64, false, 4, 64);
    run("model.4 back (C=32) x64", 32, 2, 64, 52, 52, 64, false, 4, 64);
    run("model.15 (C=32, 192->64 dual) x64", 32, 3, 64, 52, 52, 192, true, 3, 64);
    return 0;
}
