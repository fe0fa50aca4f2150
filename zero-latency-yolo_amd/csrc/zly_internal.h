// zly_internal.h -- shared declarations between the engine (engine.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "zly.h"

namespace zly {

typedef __bf16 bf16_t;

// One frame as the preprocess/decode kernels see it.
struct FrameDesc {
    unsigned long long src_off;   // byte offset of the frame's first pixel in the source buffer
    int w, h;                     // request width / height (REQUEST dims: used for box normalisation too)
};

// Implicit-GEMM convolution launch arguments.  All tensors are NHWC, batch-major; a tensor
// "view" is a channel slice [co, co+C) of a buffer whose pixels are `cs` elements apart, which is
// how C2f split / Concat cost nothing.
struct ConvArgs {
    // Field ORDER matters on the latency path: kernel arguments are fetched by scalar loads where they are first used, adjacent fields by one wide load, and
    // every further round (s_load ... s_waitcnt) in front of a ~3 us launch's first global load is ~0.1 us, thirty times per batch-1 step (measured: the two
    // reciprocals at the END of this struct cost the step 2.4 %, next to Wo / Ho they gain 1 %).  So: what the index arithmetic and the first loads need
    // comes first, in the order of use; what only the epilogue needs comes last.
    const void* in;               // input tensor (NHWC)
    // 1x1 convs only: fused "nearest-2x Upsample + Concat" input (yolov8.yaml layers 10-11, 13-14).  When in2 is
    // set, input channels [0, split_c) are read from `in`, a tensor of HALF the spatial size, at (y>>1, x>>1),
    // and channels [split_c, Cin) from `in2` at full size.  split_c is a multiple of the k-step.
    const void* in2;
    const void* wgt;              // tiled [CoutPad/16][Kpad/KSTEP][16][KSTEP]
    const float* bias;            // [CoutPad] (requested before the k-loop: with the hot fields)
    int M;                        // batch*Ho*Wo
    int cout_pad;                 // output channels the weight tiles cover (multiple of 16, >= Cout; may include whole zero tiles so that the tile count suits a kernel's channel blocking); read first by the multi-conv kernel's extent test
    int nk;                       // Kpad / KSTEP
    int Ho, Wo;
    float inv_wo, inv_ho;         // 1 / Wo, 1 / Ho: set by launch_conv / launch_conv_multi for the split-K kernel (index arithmetic by reciprocal multiplies)
    int stride, pad;
    int H, W;                     // input spatial size
    int in_cs, in_co;
    int Cin;                      // input channels as stored (multiple of 8)
    int in2_cs, in2_co, split_c;
    int K;                        // ks*ks*Cin
    // ---- epilogue ----
    void* out;        int out_cs, out_co;
    int Cout;
    const void* res;  int res_cs, res_co;    // optional residual (added after the activation)
    int act;                      // 1 = SiLU
    int out_f32;                  // 1 = write fp32 regardless of the activation dtype
};

struct ConvLaunch { int ks, ct, pt, fastk, ksplit, lds, stream, wres, ps, ws1, rowt, tpw1; };   // rowt: weight-stationary 3x3 kernel's row-tile form (ZLY_WS_ROWT, read when the shape is picked)
struct ConvArgsMulti { ConvArgs a[6]; int n; };       // independent convs of one launch (conv_igemm_multi_kernel)
hipError_t launch_conv_multi(const ConvArgsMulti& m, int ct, hipStream_t s);

// compute units the engine's streams may use: 256 (whole chip), or the size of its CU partition (engine.cpp: ZLY_CU_PART).  The
// persistent grids of the conv kernels are sized from it (a persistent workgroup that waits for a slot runs a whole round alone).
int  num_cus();
void set_num_cus(int n);

// kernels_conv.hip
hipError_t launch_conv(int dtype, const ConvArgs& a, const ConvLaunch& cfg, hipStream_t s);
void       conv_pick_config(int dtype, int ks, int stride, int cin, int cout_pad, int n, int Ho, int Wo, ConvLaunch* cfg,
                            bool streamable = false,    // single source, no residual, SiLU, bf16 output, Cout % 32 == 0
                            bool plain = false,         // single source, activation dtype output: may take the weight-stationary 3x3 kernel
                            bool dual = false);         // 1x1 with the fused Upsample + Concat input, otherwise as `streamable`: may take the weight-stationary 1x1 kernel
hipError_t conv_init();
// the 80 -> 80 class-branch convs: weight-stationary with K packed across taps (kernels_conv.hip: conv3x3_wsk_kernel); weights tiled with cin_store = 80
bool       conv_wsk_ok(int cin, int cout, int n, int Ho, int Wo);
hipError_t launch_conv_wsk(const ConvArgs& a, hipStream_t s);
int        conv_kstep(int dtype);

// kernels_pair.hip -- a C2f bottleneck (two 3x3 convs, c -> c -> c, optional shortcut) as one kernel; bf16, c = 16 / 32
struct PairArgs {
    const void* in;  int in_cs, in_co;        // x (NHWC view); also the shortcut source
    void* out;       int out_cs, out_co;
    const void* wA;  const float* bA;         // first conv: weights tiled [c/16][9][lane][frag] (frag = 8 bf16, or 4 for c = 16)
    const void* wB;  const float* bB;         // second conv
    int H, W, n;
    int TH, TW, tiles_x, tiles_y, total_tiles;
    int res;                                  // 1 = add x to the output (after the activation)
};
struct PairPlan { int th, tw, tiles_x, tiles_y, total_tiles, grid, lds_bytes; };
bool       pair_plan(int c, int n, int H, int W, PairPlan* plan);
hipError_t pair_init();
hipError_t launch_pair(int c, const PairArgs& a, const PairPlan& plan, hipStream_t s);

// a C2f block around one bottleneck as one kernel (kernels_pair.hip: c2f_kernel); mode bit 0 = cv1 in front, bit 1 = cv2 behind
struct C2fArgs {
    const void* x;  int x_cs, x_co;           // cv1 input (NHWC view); with x2 set: the half-size tensor of a fused Upsample+Concat
    const void* x2; int x2_cs, x2_co, split_c;
    const void* w1; const float* b1; int nk1; // cv1: tiled [2C/16][nk1][lane][8] (k-steps of 32); C = 16: rows in channel order, C = 32: pair-permuted
    void* cat; int cat_cs;                    // the C2f's concat buffer in HBM: [y0 | y1 | y2 | ...], C channels each
    int pair_in_co, pair_out_co;              // channel offsets of the bottleneck's input / output inside cat
    const void* wA; const float* bA; const void* wB; const float* bB; int res;     // the bottleneck (as PairArgs)
    const void* w2; const float* b2; int nk2, Cout2;   // cv2: tiled [Cout2/16][nk2][lane][frag], k-steps of C in concat order, pair-permuted rows
    void* out; int out_cs, out_co;
    int H, W, n;
    int TH, TW, tiles_x, tiles_y, total_tiles;
    int dump;                                 // also write the intermediates that would stay in LDS to cat (debug taps)
    void* mid; int mid_cs;                    // c = 64 with dump: the bottleneck's intermediate map (its first conv's output buffer of the unfused path)
};
struct C2fPlan { int th, tw, tiles_x, tiles_y, total_tiles, grid, lds_bytes, nw; };      // nw: waves per workgroup the plan was made for
bool       c2f_plan(int c, int mode, int nk1, int nk2, int cout2, int n, int H, int W, C2fPlan* plan);
hipError_t c2f_init();
hipError_t launch_c2f(int c, int mode, const C2fArgs& a, const C2fPlan& plan, hipStream_t s);
// kernels_c2f64.hip -- the same for c = 64 (c2f64_kernel: 3x3 weights in registers, 1x1 weights streamed through LDS); reached through c2f_plan / launch_c2f
bool       c2f64_plan(int mode, int nk1, int nk2, int cout2, int n, int H, int W, C2fPlan* plan);
hipError_t c2f64_init();
hipError_t launch_c2f64(int mode, const C2fArgs& a, const C2fPlan& plan, hipStream_t s);

// kernels_misc.hip
hipError_t launch_preprocess(int dtype, const uint8_t* src, const FrameDesc* desc, int n,
                             void* out_nhwc8, float* out_nchw_f32, int tw, int th, hipStream_t s);
hipError_t launch_nchw_to_nhwc8(int dtype, const float* in_nchw, void* out_nhwc8, int n, int tw, int th, hipStream_t s);
hipError_t launch_sppf_pool(int dtype, void* buf, int cs, int c, int n, int H, int W, hipStream_t s, int six_pass = 0);      // six_pass: sppf_pool_kernel also on maps of <= 16 x 16 pixels (tests / A-B)
hipError_t launch_tap_to_nchw(int dtype, const void* in, int cs, int co, int C, int H, int W, int idx, float* out, hipStream_t s);

// kernels_stem.hip -- preprocess fused into the stem conv (bf16, 16-channel stem)
struct StemArgs {
    const uint8_t* src; const FrameDesc* desc;
    const void* wgt; const float* bias;       // stem weights tiled with cin_store = 4 (k = tap*4 + c), 2 k-steps
    void* out; int out_cs, out_co;
    int tw, th, Ho, Wo, Cout, tiles_x;
};
hipError_t launch_stem_fused(const StemArgs& a, int n, hipStream_t s);
int stem_tiles_x(int Wo);
// preprocess + model.0 + model.1 in one kernel (the stem map stays in LDS); st.out is only written with dump = 1 (debug taps)
struct Stem1Args {
    StemArgs st;
    const void* w1; const float* b1;          // model.1 weights tiled [2][9 taps][lane][4] (k = ci per tap, pair-permuted rows), bias in channel order
    void* out1; int out1_cs, out1_co;
    int H1, W1;                               // model.1 output map
    int TH, TW, tiles_x, tiles_y;
    int dump;
    int nw, var;                              // waves per workgroup (0 = default), kernel variant (2 = persistent workgroups + input prefetch, 1 = one tile per workgroup, 0 = round 3's staging / tap order)
    int n;                                    // frames (set by launch_stem_model1)
    int pgrid;                                // var 2: workgroups of the persistent grid (0 = as many as stay resident)
    int small_tiles;                          // 1: launch_stem_model1 halves the tile for launches with fewer workgroups than the chip holds (small batches)
    // set by launch_stem_model1 (host IEEE divides: the kernel used to spend six fp32 divides per thread on them)
    float inv_pw, inv_rw, inv_tw, inv_qb;     // 1 / patch row pitch, 1 / region width, 1 / tile width, 1 / quad blocks per patch row
    const void* wgt0p;                        // stem weights in the tap order of the conflict-free fragment reads (kernels_stem.hip: STEM1_TAP_SLOT)
};
void       stem1_plan(int H1, int W1, int* th, int* tw);
hipError_t stem1_init();
hipError_t launch_stem_model1(const Stem1Args& a, int n, hipStream_t s);
const int* stem1_tap_slot();              // [9]: k slot of tap ky * 3 + kx in Stem1Args::wgt0p (weights.h: repack_conv's tap_slot)

// kernels_sppf.hip -- SPPF (cv1 -> three 5x5 max pools -> cv2 over the concat) as one kernel; bf16, hidden width 128, maps of up to 176 pixels
struct SppfArgs {
    const void* x; int x_cs, x_co, Cin;       // cv1 input (NHWC view)
    const void* w1; const float* b1;          // cv1: tiled [c/16][Cin/32][lane][8], pair-permuted rows
    const void* w2; const float* b2;          // cv2: tiled [Cout/16][4c/32][lane][8], k in concat order [y | p1 | p2 | p3], pair-permuted rows
    void* out; int out_cs, out_co, Cout;
    void* cat; int cat_cs;                    // the block's concat buffer in HBM: written only with dump (debug taps)
    int H, W, n, c;
    int split;                                // workgroups per frame (2 / 4), 0 = chosen from n
    int dump;
};
bool       sppf_fused_ok(int cin, int c, int cout, int H, int W);
int        sppf_split(int cout, int n);
hipError_t sppf_init();
hipError_t launch_sppf_fused(const SppfArgs& a, hipStream_t s);

// kernels_head.hip -- fused Detect head (final 1x1 convs + DFL + dist2bbox + sigmoid + decode/threshold)
struct HeadLevel {
    const void* box_in; const void* cls_in;   // [n][H*W][cs] activations of the two branches' second 3x3 convs
    int box_cs, cls_cs, box_cin, cls_cin;
    const void* wb; const void* wc;           // final 1x1 weights, tiled in MFMA lane order
    const float* bb; const float* bc;         // biases (padded to 16)
    int nkb, nkc;                             // k-steps of the two GEMMs
    int H, W, hw, stride_px, anchor_off, block0;
    float* logits; int logits_cs;             // optional fp32 [n][H*W][logits_cs] dump (debug taps), or null
};
#define HEAD_KMAX 8                           // most k-steps of one Detect branch the fused tail holds in registers (bf16: 256 channels, fp32: 128)
#define HEAD_WAVES 8                          // waves per workgroup of the fused Detect tail, one 16-anchor tile each
#define HEAD_GROUP (HEAD_WAVES * 16)          // anchors per workgroup; HeadLevel::block0 / total_blocks count these groups
struct HeadArgs {
    HeadLevel lv[3];
    int nc, N_total, total_blocks;
    int only_level;                           // -1: all three levels in one launch; 0..2: that level only (its own launch, beside the neck)
    float* head;                              // [n][4+nc][N_total] or null
    const FrameDesc* desc; float conf_thr;
    float skip_logit;                         // set by launch_head_fused: class logit below which no score reaches conf_thr
    int diag;                                 // diagnostic builds only (tools/head_bench.hip); 0 in the product
    struct Cand* cand; int* cand_count;
    int buf32;                                // set by launch_head_fused: every level's branch tensors are below 2 GiB -> fragments by buffer loads (lane mask as an out-of-range offset)
};
hipError_t launch_head_fused(int dtype, const HeadArgs& a, int n, hipStream_t s);

// kernels_post.hip
struct Cand { float x, y, w, h; float conf; int cls; int anchor; int pad_; };   // 32 bytes
hipError_t launch_decode(const float* head, int nc, int N, int n, const FrameDesc* desc, float conf_thr,
                         Cand* cand, int* cand_count, hipStream_t s);
hipError_t nms_init();
hipError_t launch_nms(const Cand* cand, int* cand_count, int N, int n, float iou_thr, int nc,
                      Cand* scratch, void* slabs, int cap, uint32_t tag0, hipStream_t s, int force_general = 0);

}  // namespace zly
