// diag_lds.hip -- DIAGNOSTIC build of the LDS-tiled conv kernel with s_memtime stamps around the phases of an
// item (never part of libzly.so).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Izero-latency-yolo_amd/csrc -DZLY_DIAG=1 \
//         zero-latency-yolo_amd/tools/diag_lds.hip -o zero-latency-yolo_amd/_build/diag_lds && ./zero-latency-yolo_amd/_build/diag_lds
#include "../csrc/kernels_conv.hip"
#include <stdio.h>
#include <vector>
#include <string.h>
using namespace zly;

template <int S, int CT, int PT>
static void run(const char* name, int n, int H, int W, int Cin, int Cout)
{
    const int Ho = H / S, Wo = W / S;
    const int nk = 9 * Cin / 32, cout_pad = (Cout + 15) / 16 * 16;
    std::vector<uint16_t> hin((size_t)n * H * W * Cin), hw((size_t)cout_pad * nk * 32);
    for (size_t i = 0; i < hin.size(); ++i) hin[i] = 0x3c00 + (uint16_t)((i * 2654435761u >> 20) & 0x1ff);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3800 + (uint16_t)((i * 40503u >> 7) & 0x3ff) ^ ((i & 1) << 15);
    void *din, *dw, *dout; float* dbias; unsigned long long* ddbg;
    hipMalloc(&din, hin.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dout, (size_t)n * Ho * Wo * Cout * 2);
    hipMalloc((void**)&dbias, cout_pad * 4); hipMemset(dbias, 0, cout_pad * 4);
    hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    ConvArgs a; memset(&a, 0, sizeof a);
    a.in = din; a.in_cs = Cin; a.in_co = 0; a.H = H; a.W = W; a.Cin = Cin; a.wgt = dw; a.bias = dbias;
    a.out = dout; a.out_cs = Cout; a.out_co = 0; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.stride = S; a.pad = 1;
    a.K = 9 * Cin; a.nk = nk; a.M = n * Ho * Wo; a.act = 1;
    const int ct = CT, ytiles = cout_pad / (16 * ct), th = 4 * PT;
    const int tiles_x = (Wo + 15) / 16, tiles_y = (Ho + th - 1) / th, tpi = tiles_x * tiles_y, total = tpi * n;
    int gx = total; if (gx * ytiles > 512) gx = 512 / ytiles;
    const size_t nw = (size_t)gx * ytiles * 4;
    hipMalloc((void**)&ddbg, nw * 64); hipMemset(ddbg, 0, nw * 64);
    a.in2 = ddbg;
    conv_init();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((conv3x3_lds_kernel<S, CT, PT>), dim3(gx, ytiles), dim3(256), lds_bytes(S, PT, CT), 0, a, tiles_x, tpi, total);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
    }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nw * 8);
    hipMemcpy(h.data(), ddbg, nw * 64, hipMemcpyDeviceToHost);
    double s[8] = {0}; for (size_t w = 0; w < nw; ++w) for (int k = 0; k < 8; ++k) s[k] += (double)h[w * 8 + k];
    const double items = s[5] / nw;
    printf("%-28s grid %dx%d  %.1f us  items/wave %.1f  cycles/item: store+vmwait %.0f  barrier1 %.0f  load-issue %.0f  taps %.0f  epilogue %.0f  barrier2 %.0f | kernel body %.0f cycles/wave (%.0f per item)\n",
           name, gx, ytiles, ms * 1e3, items, s[0] / s[5], s[1] / s[5], s[2] / s[5], s[3] / s[5], s[7] / s[5], s[4] / s[5], s[6] / nw, s[6] / s[5]);
    hipFree(din); hipFree(dw); hipFree(dout); hipFree(dbias); hipFree(ddbg);
}

int main()
{
    run<1, 4, 2>("26x26 64->64 (x64)", 64, 26, 26, 64, 64);
    run<1, 4, 1>("13x13 128->128 (x64)", 64, 13, 13, 128, 128);
    run<1, 3, 2>("52x52 64->144 (x64) P3 stem", 64, 52, 52, 64, 144);
    run<1, 4, 2>("52x52 64->64 (x64) box2", 64, 52, 52, 64, 64);
    run<1, 2, 2>("52x52 32->32 (x64)", 64, 52, 52, 32, 32);
    run<2, 4, 1>("104->52 s2 32->64 (x64)", 64, 104, 104, 32, 64);
    run<2, 4, 1>("52->26 s2 64->128 (x64)", 64, 52, 52, 64, 128);
    run<2, 4, 1>("26->13 s2 128->256 (x64)", 64, 26, 26, 128, 256);
    return 0;
}
