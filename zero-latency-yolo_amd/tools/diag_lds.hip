// diag_lds.hip -- DIAGNOSTIC build of the LDS-tiled conv kernel with s_memtime stamps around the phases of an
// item (never part of libzly.so).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Izero-latency-yolo_amd/csrc -DZLY_DIAG=1 \
//         zero-latency-yolo_amd/tools/diag_lds.hip -o zero-latency-yolo_amd/_build/diag_lds && ./zero-latency-yolo_amd/_build/diag_lds
#include "../csrc/kernels_conv.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <string.h>
using namespace zly;

template <int S, int CT, int PT>
static void run(const char* name, int n, int H, int W, int Cin, int Cout, int wres = 0)
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
    hipMalloc((void**)&ddbg, nw * 128); hipMemset(ddbg, 0, nw * 128);
    a.in2 = ddbg;
    conv_init();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((conv3x3_lds_kernel<S, CT, PT>), dim3(gx, ytiles), dim3(256), lds_bytes(S, PT, CT, wres ? Cin / 32 : 1), 0, a, tiles_x, tpi, total, wres);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
    }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nw * 16);
    hipMemcpy(h.data(), ddbg, nw * 128, hipMemcpyDeviceToHost);
    double s[8] = {0}; for (size_t w = 0; w < nw; ++w) for (int k = 0; k < 8; ++k) s[k] += (double)h[w * 16 + k];
    // wall clock (s_memrealtime, 100 MHz): first body start .. last body end over all waves, mean body length, and how
    // many waves were inside their body at the middle of that span (= resident concurrency)
    unsigned long long r0 = ~0ull, r1 = 0; double rsum = 0;
    for (size_t w = 0; w < nw; ++w) { r0 = std::min(r0, h[w * 16 + 8]); r1 = std::max(r1, h[w * 16 + 9]); rsum += (double)(h[w * 16 + 9] - h[w * 16 + 8]); }
    const unsigned long long mid = r0 + (r1 - r0) / 2; size_t live = 0;
    for (size_t w = 0; w < nw; ++w) live += h[w * 16 + 8] <= mid && mid < h[w * 16 + 9];
    printf("    wall: all bodies span %.2f us, mean body %.2f us (%.0f ticks/us), waves in their body at mid-span: %zu of %zu\n",
           (r1 - r0) / 100.0, rsum / nw / 100.0, (s[6] / nw) / (rsum / nw / 100.0), live, nw);
    const double items = s[5] / nw;
    printf("%-28s grid %dx%d  %.1f us  items/wave %.1f  cycles/item: store+vmwait %.0f  barrier1 %.0f  load-issue %.0f  taps %.0f  epilogue %.0f  barrier2 %.0f | kernel body %.0f cycles/wave (%.0f per item)\n",
           name, gx, ytiles, ms * 1e3, items, s[0] / s[5], s[1] / s[5], s[2] / s[5], s[3] / s[5], s[7] / s[5], s[4] / s[5], s[6] / nw, s[6] / s[5]);
    hipFree(din); hipFree(dw); hipFree(dout); hipFree(dbias); hipFree(ddbg);
}

// calibration: what one s_memtime tick is (wall clock) and how many ticks a v_mfma_f32_16x16x32_bf16 takes when a wave
// issues them back to back on 4 independent accumulators (one wave per SIMD, every CU busy)
__global__ __launch_bounds__(256) void calib_kernel(unsigned long long* out, int iters)
{
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (bf16_t)(float)(threadIdx.x + j); b[j] = (bf16_t)(float)(j + 1); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {      // inline asm: left to itself hipcc rotates the accumulators through AGPR copies here
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t"
                     "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %5, %3"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (unsigned long long)(c0[0] + c1[1] + c2[2] + c3[3]); }
}

static void calibrate()
{
    unsigned long long* d; hipMalloc((void**)&d, 1024 * 16);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[4] = {1, 64, 256, 512};
    for (int gi = 0; gi < 4; ++gi) {
        const int g = grids[gi];
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(calib_kernel, dim3(g), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long h[1024]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        double ticks = 0, tmin = 1e30, tmax = 0;
        for (int i = 0; i < g; ++i) { const double t = (double)h[i * 2]; ticks += t; if (t < tmin) tmin = t; if (t > tmax) tmax = t; }
        ticks /= g;
        printf("calibration: %d workgroups x 4 waves, %d x 4 MFMA 16x16x32 bf16 per wave: %.1f us wall, s_memtime ticks/wave mean %.0f (min %.0f max %.0f) -> %.1f ticks/us, %.2f ticks per MFMA (min %.2f), %.1f TFLOP/s\n",
               g, iters, ms * 1e3, ticks, tmin, tmax, ticks / (ms * 1e3), ticks / (iters * 4.0), tmin / (iters * 4.0), (double)g * 4 * iters * 4 * 16384.0 / (ms * 1e-3) * 1e-12);
    }
    hipFree(d);
}

int main()
{
    calibrate();
    run<1, 4, 2>("26x26 64->64 (x64)", 64, 26, 26, 64, 64);
    run<1, 4, 1>("13x13 128->128 (x64)", 64, 13, 13, 128, 128);
    run<1, 3, 2>("52x52 64->144 (x64) P3 stem", 64, 52, 52, 64, 144);
    run<1, 4, 2>("52x52 64->64 (x64) box2", 64, 52, 52, 64, 64);
    run<1, 2, 2>("52x52 32->32 (x64)", 64, 52, 52, 32, 32);
    run<1, 2, 2>("52x52 32->32 (x64) resident w", 64, 52, 52, 32, 32, 1);
    run<1, 2, 2>("52x52 64->64 box2 CT=2 resident w", 64, 52, 52, 64, 64, 1);
    run<1, 2, 2>("26x26 64->64 CT=2 resident w", 64, 26, 26, 64, 64, 1);
    run<1, 3, 2>("52x52 64->144 P3 stem resident w", 64, 52, 52, 64, 144, 1);
    run<2, 4, 1>("104->52 s2 32->64 (x64)", 64, 104, 104, 32, 64);
    run<2, 4, 1>("52->26 s2 64->128 (x64)", 64, 52, 52, 64, 128);
    run<2, 4, 1>("26->13 s2 128->256 (x64)", 64, 26, 26, 128, 256);
    return 0;
}
