// head_bench.hip -- stand-alone timing of head_fused_kernel (the ablation of its predecessor is in profiles/r03_head_kernel_ablation.txt).
// Originally: DIAGNOSTIC: what binds head_fused_kernel?  Runs the Detect tail of a YOLOv8n 416x416 batch (3549 anchors x n frames,
// random bf16 activations, class logits ~N(-2.75, 0.7) so that ~1.5 % of the anchors pass conf 0.5) with parts switched off:
//   bit 0: no weight-fragment loads (L1 / L2 traffic), bit 1: no activation loads (HBM), bit 2: stop after the class GEMM (no epilogue).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Izero-latency-yolo_amd/csrc -DZLY_HEAD_DIAG=1 zero-latency-yolo_amd/tools/head_bench.hip \
//         -o zero-latency-yolo_amd/_build/head_bench && ./zero-latency-yolo_amd/_build/head_bench
#include "../csrc/kernels_head.hip"
#include <stdio.h>
#include <string.h>
#include <random>
#include <vector>
using namespace zly;

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 64;
    const int hw[3] = {2704, 676, 169}, Ws[3] = {52, 26, 13};
    std::mt19937 rng(3);
    std::normal_distribution<float> nd(0.f, 1.f);
    HeadArgs a; memset(&a, 0, sizeof a);
    int block0 = 0, off = 0;
    for (int l = 0; l < 3; ++l) {
        HeadLevel& L = a.lv[l];
        std::vector<uint16_t> hb((size_t)n * hw[l] * 64), hc((size_t)n * hw[l] * 80), wb(4 * 2 * 512), wc(5 * 3 * 512);
        for (auto& v : hb) v = f2bf(0.25f * nd(rng));
        for (auto& v : hc) v = f2bf(0.25f * nd(rng));
        for (auto& v : wb) v = f2bf(0.3f * nd(rng));
        for (auto& v : wc) v = f2bf(0.31f * nd(rng));            // logit std ~0.7 over 80 inputs of std 0.25
        std::vector<float> bb(64, 1.0f), bc(80, -2.75f);
        void *dhb, *dhc, *dwb, *dwc; float *dbb, *dbc;
        hipMalloc(&dhb, hb.size() * 2); hipMalloc(&dhc, hc.size() * 2); hipMalloc(&dwb, wb.size() * 2); hipMalloc(&dwc, wc.size() * 2);
        hipMalloc((void**)&dbb, 64 * 4); hipMalloc((void**)&dbc, 80 * 4);
        hipMemcpy(dhb, hb.data(), hb.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dhc, hc.data(), hc.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dwb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dwc, wc.data(), wc.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dbb, bb.data(), 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dbc, bc.data(), 80 * 4, hipMemcpyHostToDevice);
        L.box_in = dhb; L.cls_in = dhc; L.box_cs = 64; L.cls_cs = 80; L.box_cin = 64; L.cls_cin = 80;
        L.wb = dwb; L.wc = dwc; L.bb = dbb; L.bc = dbc; L.nkb = 2; L.nkc = 3;
        L.H = Ws[l]; L.W = Ws[l]; L.hw = hw[l]; L.stride_px = 8 << l; L.anchor_off = off; L.block0 = block0; L.logits = nullptr; L.logits_cs = 144;
        block0 += (hw[l] + HEAD_GROUP - 1) / HEAD_GROUP; off += hw[l];
    }
    a.nc = 80; a.N_total = off; a.total_blocks = block0; a.only_level = -1; a.head = nullptr; a.conf_thr = 0.5f;
    FrameDesc* dd; std::vector<FrameDesc> hd((size_t)n); for (auto& d : hd) { d.src_off = 0; d.w = 416; d.h = 416; }
    hipMalloc((void**)&dd, n * sizeof(FrameDesc)); hipMemcpy(dd, hd.data(), n * sizeof(FrameDesc), hipMemcpyHostToDevice);
    a.desc = dd;
    hipMalloc((void**)&a.cand, (size_t)n * off * sizeof(Cand)); hipMalloc((void**)&a.cand_count, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[8] = {"full kernel", "no weight loads", "no activation loads", "no weight + no activation loads", "stop after class GEMM",
                            "stop after class GEMM, no weight loads", "stop after class GEMM, no activation loads", "stop after class GEMM, no loads at all"};
    for (int round = 0; round < 2; ++round)                          // round 0 warms the clocks up
        for (int mode = 0; mode < 1; ++mode) {       // the ablation switches (modes 1-7: profiles/r03_head_kernel_ablation.txt) lived in the one-tile-per-workgroup-wave kernel of round 2
            a.diag = mode;
            float best = 1e9f, ms = 0;
            for (int rep = 0; rep < 30; ++rep) {
                hipMemsetAsync(a.cand_count, 0, n * 4, 0);
                hipEventRecord(e0, 0);
                launch_head_fused(ZLY_DTYPE_BF16, a, n, 0);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            if (round == 1) {
                std::vector<int> cnt((size_t)n); hipMemcpy(cnt.data(), a.cand_count, n * 4, hipMemcpyDeviceToHost);
                long tot = 0; for (int c : cnt) tot += c;
                printf("mode %d %-46s %7.1f us   (%.1f candidates per frame)\n", mode, names[mode], best * 1e3, (double)tot / n);
            }
        }
    return 0;
}
