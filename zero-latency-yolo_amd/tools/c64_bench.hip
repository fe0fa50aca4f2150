// c64_bench.hip -- DIAGNOSTIC build of c2f64_kernel with s_memtime stamps around its phases (never part of libzly.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Izero-latency-yolo_amd/csrc -DZLY_C64_DIAG=1 zero-latency-yolo_amd/tools/c64_bench.hip \
//         -o zero-latency-yolo_amd/_build/c64_bench && ./zero-latency-yolo_amd/_build/c64_bench
// Per block shape (model.12 / model.18 / model.6 front + back at batch 64 and batch 1): time of the launch, and the per-wave cycle sums of
// the phases: A weights+first fragments | A rounds | B weight loads | B barrier | B loop | C weight loads | C barrier | C loop | D weights | D rounds.
#include "../csrc/kernels_c2f64.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
namespace zly { int num_cus() { return 256; } }
using namespace zly;

static void* dalloc_rand(size_t elems, unsigned seed, unsigned short base, unsigned mask)
{
    std::vector<uint16_t> h(elems);
    for (size_t i = 0; i < elems; ++i) h[i] = (uint16_t)(base + (((unsigned)i * 2654435761u + seed) >> 20 & mask)) ^ (uint16_t)((i & 1) << 15);
    void* d; hipMalloc(&d, elems * 2); hipMemcpy(d, h.data(), elems * 2, hipMemcpyHostToDevice);
    return d;
}

static void run(const char* name, int mode, int n, int H, int W, int cin, bool dual, int nmaps)
{
    const int nk1 = cin / 32, nk2 = 2 * nmaps;
    C2fPlan pl{};
    if (!c2f64_plan(mode, nk1, nk2, 128, n, H, W, &pl)) { printf("%s: no plan\n", name); return; }
    C2fArgs a; memset(&a, 0, sizeof a);
    const size_t px = (size_t)n * H * W;
    if (dual) { a.x = dalloc_rand(px / 4 * 256, 1, 0x3c00, 0x1ff); a.x_cs = 256; a.x2 = dalloc_rand(px * 128, 2, 0x3c00, 0x1ff); a.x2_cs = 128; a.split_c = 256; }
    else { a.x = dalloc_rand(px * cin, 1, 0x3c00, 0x1ff); a.x_cs = cin; }
    a.w1 = dalloc_rand((size_t)8 * (nk1 ? nk1 : 1) * 512, 3, 0x3400, 0x3ff); a.nk1 = nk1;
    a.wA = dalloc_rand(4 * 18 * 512, 4, 0x3000, 0x3ff); a.wB = dalloc_rand(4 * 18 * 512, 5, 0x3000, 0x3ff);
    a.w2 = dalloc_rand((size_t)8 * nk2 * 512, 6, 0x3000, 0x3ff); a.nk2 = nk2; a.Cout2 = 128;
    float* bias; hipMalloc((void**)&bias, 512 * 4); hipMemset(bias, 0, 512 * 4);
    a.b1 = bias; a.bA = bias; a.bB = bias; a.b2 = bias;
    a.cat = dalloc_rand(px * 64 * nmaps, 7, 0x3c00, 0x1ff); a.cat_cs = 64 * nmaps;
    a.pair_in_co = mode == 2 ? 128 : 64; a.pair_out_co = a.pair_in_co + 64; a.res = 1;
    void* out; hipMalloc(&out, px * 128 * 2); a.out = out; a.out_cs = 128;
    a.H = H; a.W = W; a.n = n; a.TH = pl.th; a.TW = pl.tw; a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.total_tiles = pl.total_tiles;
    const size_t nw = (size_t)pl.grid * 8;
    unsigned long long* ddbg; hipMalloc((void**)&ddbg, nw * 128); hipMemset(ddbg, 0, nw * 128);
    hipMemcpyToSymbol(HIP_SYMBOL(g_c64_diag), &ddbg, sizeof ddbg);
    c2f64_init();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 30; ++rep) {
        hipEventRecord(e0, 0);
        if (launch_c2f64(mode, a, pl, 0) != hipSuccess) { printf("%s: launch failed\n", name); return; }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 10 && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(nw * 16);
    hipMemcpy(h.data(), ddbg, nw * 128, hipMemcpyDeviceToHost);
    double s[11] = {0}; for (size_t w = 0; w < nw; ++w) for (int k = 0; k < 11; ++k) s[k] += (double)h[w * 16 + k];
    const double tpw = (double)pl.total_tiles / pl.grid;
    printf("%-34s tile %dx%d, %d tiles on %d workgroups, lds %d KB: %.1f us best of 20\n   cycles per tile per wave (mean): A weights+frags %.0f | A rounds %.0f | B weights %.0f | B barrier %.0f | B loop %.0f | C weights %.0f | C barrier %.0f | C loop %.0f | D weights %.0f | D rounds %.0f | wave total %.0f\n",
           name, pl.th, pl.tw, pl.total_tiles, pl.grid, pl.lds_bytes / 1024, best * 1e3,
           s[0] / nw / tpw, s[1] / nw / tpw, s[2] / nw / tpw, s[3] / nw / tpw, s[4] / nw / tpw, s[5] / nw / tpw, s[6] / nw / tpw, s[7] / nw / tpw, s[8] / nw / tpw, s[9] / nw / tpw, s[10] / nw / tpw);
}

int main()
{
    run("model.18 (192 -> 128, n=1) x64", 3, 64, 26, 26, 192, false, 3);
    run("model.12 (384 -> 128 dual) x64", 3, 64, 26, 26, 384, true, 3);
    run("model.6 front (128 -> 128) x64", 1, 64, 26, 26, 128, false, 4);
    run("model.6 back x64", 2, 64, 26, 26, 128, false, 4);
    run("model.18 x1", 3, 1, 26, 26, 192, false, 3);
    run("model.12 x1", 3, 1, 26, 26, 384, true, 3);
    return 0;
}
