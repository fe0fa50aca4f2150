// igemm_bench.hip -- DIAGNOSTIC build of the split-K conv kernel of the latency path (conv_igemm_kernel<KSPLIT = 4>) with s_memtime stamps (never part of libzly.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Izero-latency-yolo_amd/csrc -DZLY_IGEMM_DIAG=1 zero-latency-yolo_amd/tools/igemm_bench.hip \
//         -o zero-latency-yolo_amd/_build/igemm_bench && ./zero-latency-yolo_amd/_build/igemm_bench
// Batch-1 shapes of YOLOv8n at 416 x 416: a chain of 40 launches of ONE layer (ping-pong buffers) as a graph -> us per launch, and per wave the stamps
//   start | loads requested | loads arrived | MFMAs issued | barrier passed | epilogue done (wave 0) | stores acknowledged (wave 0)
// as differences in s_memtime ticks (mean over the waves of the last launch; the stamped build waits for the loads and the stores explicitly, which the
// product does not: read the SHARES).
#include "../csrc/kernels_conv.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
using namespace zly;

static void run(const char* name, int H, int W, int Cin, int Cout, int ks, int stride)
{
    const int n = 1, Ho = H / stride, Wo = W / stride, nk = ks * ks * Cin / 32, cout_pad = (Cout + 15) / 16 * 16;
    std::vector<uint16_t> hin((size_t)n * H * W * Cin), hw((size_t)cout_pad * nk * 32);
    for (size_t i = 0; i < hin.size(); ++i) hin[i] = 0x3c00 + (uint16_t)((i * 2654435761u >> 20) & 0x1ff);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3000 + (uint16_t)((i * 40503u >> 7) & 0x3ff) ^ ((i & 1) << 15);
    void *din, *dw, *dout; float* dbias; unsigned long long* ddbg;
    (void)hipMalloc(&din, hin.size() * 2); (void)hipMalloc(&dw, hw.size() * 2); (void)hipMalloc(&dout, (size_t)n * Ho * Wo * cout_pad * 2);
    (void)hipMalloc((void**)&dbias, cout_pad * 4); (void)hipMemset(dbias, 0, cout_pad * 4);
    (void)hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    ConvArgs a; memset(&a, 0, sizeof a);
    a.in = din; a.in_cs = Cin; a.H = H; a.W = W; a.Cin = Cin; a.wgt = dw; a.bias = dbias;
    a.out = dout; a.out_cs = cout_pad; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.cout_pad = cout_pad; a.stride = stride; a.pad = ks / 2;
    a.K = ks * ks * Cin; a.nk = nk; a.M = n * Ho * Wo; a.act = 1;
    ConvLaunch cfg{};
    conv_pick_direct(ZLY_DTYPE_BF16, ks, Cin, cout_pad, a.M, &cfg);
    if (cfg.ksplit != 4) { printf("%-34s not a split-K shape (ct %d pt %d)\n", name, cfg.ct, cfg.pt); return; }
    const int gx = (a.M + 15) / 16, gy = cout_pad / (16 * cfg.ct);
    const size_t nw = (size_t)gx * gy * 4;
    (void)hipMalloc((void**)&ddbg, nw * 64); (void)hipMemset(ddbg, 0, nw * 64);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_igemm_diag), &ddbg, sizeof ddbg);
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int N = 40;
    hipGraph_t gr; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) if (launch_conv(ZLY_DTYPE_BF16, a, cfg, st) != hipSuccess) { printf("%s: launch failed\n", name); return; }
    (void)hipStreamEndCapture(st, &gr);
    (void)hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f, ms;
    for (int rep = 0; rep < 20; ++rep) {
        (void)hipEventRecord(e0, st); (void)hipGraphLaunch(ge, st); (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(nw * 8);
    (void)hipMemcpy(h.data(), ddbg, nw * 64, hipMemcpyDeviceToHost);
    double d[6] = {0}; size_t n0 = 0, nall = 0;
    for (size_t w = 0; w < nw; ++w) {
        const unsigned long long* o = &h[w * 8];
        if (!o[0]) continue;
        ++nall;
        for (int k = 0; k < 4; ++k) d[k] += (double)(o[k + 1] - o[k]);
        if ((w & 3) == 0) { ++n0; d[4] += (double)(o[5] - o[4]); d[5] += (double)(o[6] - o[5]); }
    }
    printf("%-34s grid %3d x %2d (CT %d), %2d k-steps per wave: %5.2f us per launch | ticks per wave: index math + requests %5.0f | loads in flight %5.0f | MFMAs %5.0f | LDS combine + barrier %5.0f | epilogue (wave 0) %5.0f | stores acknowledged %5.0f\n",
           name, gx, gy, cfg.ct, (nk + 3) / 4, best * 1e3 / N, d[0] / nall, d[1] / nall, d[2] / nall, d[3] / nall, n0 ? d[4] / n0 : 0.0, n0 ? d[5] / n0 : 0.0);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(gr); (void)hipStreamDestroy(st);
    (void)hipFree(din); (void)hipFree(dw); (void)hipFree(dout); (void)hipFree(dbias); (void)hipFree(ddbg);
}

int main()
{
    (void)conv_init();
    printf("# ticks = s_memtime (about the shader clock here: the columns of a row add up to the in-kernel part of its launch time)\n");
    run("26x26 64->64 3x3 (model.6.m)", 26, 26, 64, 64, 3, 1);
    run("26x26 128->128 1x1 (model.6.cv1)", 26, 26, 128, 128, 1, 1);
    run("13x13 128->128 3x3 (model.8.m)", 13, 13, 128, 128, 3, 1);
    run("13x13 256->256 1x1 (model.8.cv1)", 13, 13, 256, 256, 1, 1);
    run("13x13 512->256 1x1 (model.9.cv2)", 13, 13, 512, 256, 1, 1);
    run("26x26 128->256 3x3 s2 (model.7)", 26, 26, 128, 256, 3, 2);
    run("52x52 64->128 3x3 s2 (model.5)", 52, 52, 64, 128, 3, 2);
    return 0;
}
