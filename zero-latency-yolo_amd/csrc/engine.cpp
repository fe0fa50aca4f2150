// engine.cpp -- the detect engine behind the C ABI of include/zly.h: model plan, HBM layout,
// stream/graph orchestration.  Compiled with hipcc; the kernels live in kernels_*.hip.
//
// Path (reference OnnxInferenceEngine::runInference, src/inference/onnx_engine.cpp:518-646):
//   preprocess -> YOLOv8 forward (57 convs as 54 MFMA launches, SPPF pools, 2 upsamples) -> fused Detect
//   tail (6 final 1x1 convs + DFL + sigmoid + decode/threshold) -> class-aware NMS -> result slab per frame.
//
// HBM layout: every activation is NHWC, batch-major, in the engine dtype (bf16 or fp32).  Concat
// and C2f's split never move data: producers write into channel slices of the consumer's concat
// buffer (ConvArgs::out_cs/out_co) and consumers read channel slices (in_cs/in_co).
#include "zly_internal.h"
#include "weights.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace zly {

static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) { g_last_error = msg; return code; }

#define HIP_TRY(expr, code)                                                                        \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail((code), std::string(#expr) + ": " + hipGetErrorString(_e));                \
    } while (0)

struct View { int buf; int co; int C; };

struct Buffer {
    std::string name;
    int H = 0, W = 0, C = 0;
    bool f32 = false;          // always fp32 (final Detect logits) regardless of engine dtype
    void* ptr = nullptr;
    size_t elems_per_frame() const { return (size_t)H * W * C; }
};

enum OpKind { OP_PREPROCESS = 0, OP_CONV = 1, OP_SPPF = 2, OP_HEAD = 4, OP_NMS = 6 };

struct Op {
    int kind = 0;
    std::string name;
    // conv
    View in{-1, 0, 0}, out{-1, 0, 0}, res{-1, 0, 0};
    View in2{-1, 0, 0};                // 1x1 convs: second (full-size) source of a fused Upsample+Concat input; `in` is then half-size
    int ks = 1, stride = 1, act = 0, out_f32 = 0;
    int cout = 0, cout_pad = 0, nk = 0, K = 0;
    size_t w_off = 0, b_off = 0;       // offsets into the device weight blob
    std::vector<std::string> taps;     // conv names whose outputs this op produces (for zly_debug_tap)
    std::vector<int> tap_co;           // channel offset of each tap inside `out`
    std::vector<int> tap_c;
    // sppf / upsample / head
    int c = 0;                         // sppf: hidden width; upsample: channels
    int level = 0, stride_px = 0, anchor_off = 0;
    int lane = 0;                      // 0 = main stream; 1, 2 = Detect-branch streams that run beside the neck
    // fused bottleneck pair (kernels_pair.hip): pair = 1 on a bottleneck's first 3x3 conv (it then carries what the
    // kernel needs of the second one), 2 on the second (a no-op when the pair kernel ran); 0 otherwise
    int pair = 0, pair_c = 0, pair_res = 0;
    View pair_out{-1, 0, 0};
    size_t pair_wA = 0, pair_bA = 0, pair_wB = 0, pair_bB = 0;
    // fused C2f block (kernels_pair.hip: c2f_kernel): c2f_mode != 0 on the op that launches it (cv1, or the second bottleneck's first
    // conv for the back half of a two-bottleneck C2f); the ops it covers carry the leader's name and launch nothing when it is active
    int c2f_mode = 0, c2f_c = 0, c2f_res = 0, c2f_cat = -1, c2f_in_co = 0, c2f_out_co = 0, c2f_nk1 = 0, c2f_nk2 = 0, c2f_cout2 = 0;
    View c2f_x{-1, 0, 0}, c2f_x2{-1, 0, 0}, c2f_out{-1, 0, 0}, c2f_mid{-1, 0, 0};
    size_t c2f_w1 = 0, c2f_b1 = 0, c2f_wA = 0, c2f_bA = 0, c2f_wB = 0, c2f_bB = 0, c2f_w2 = 0, c2f_b2 = 0;
    std::string c2f_leader_name;       // covered ops (and the leader itself)
    int c2f_leader = -1;               // index of the leader op (resolved after the ops are ordered)
    int c2f_vis = 0;                   // this op's output with the fused kernel: 0 = in HBM, 1 = only with ZLY_FLAG_DUMP_LOGITS, 2 = stays in LDS
    HeadArgs head{};                   // OP_HEAD: fused Detect tail
    int head_box[3] = {-1, -1, -1}, head_cls[3] = {-1, -1, -1};     // OP_HEAD: buffers the tail reads per level (launch-group bookkeeping)
    double flops = 0, bytes = 0;
    double wbytes = 0;                 // weight + bias bytes of this op (part of `bytes`)
    double cin_frac = 1.0;             // real / stored input channels (the 3-of-8 channel image, the 80-of-96 channel class branch)
    size_t wsk_w = 0;                  // the 80 -> 80 class-branch convs: weights once more, tiled with cin_store = 80 for conv3x3_wsk_kernel (K packed across taps); 0 = none
    std::map<int, int> wsk_cache;      // batch size -> does that kernel take the launch
    std::map<int, ConvLaunch> launch_cache;    // batch size -> kernel shape of the per-conv path (conv_pick_config reads the environment: once per shape, not per launch)
};

struct Ingest;        // pipelined host-to-host path (zly_submit / zly_wait), below

}  // namespace zly

using namespace zly;

struct zly_engine {
    zly_config cfg;
    std::string weights_path;
    int dev = 0;
    int dtype = ZLY_DTYPE_BF16;
    size_t esz = 2;
    ModelFile model;
    int nc = 0, N = 0;
    int lvl_h[3], lvl_w[3];
    std::vector<Buffer> bufs;
    std::vector<Op> ops;
    int in_buf = -1;
    std::map<std::string, std::pair<int, int>> tap_index;   // conv name -> (op index, tap slot)
    std::map<std::string, std::pair<int, int>> tap_final;   // final Detect convs -> (logits buffer, channel offset)

    void* d_weights = nullptr;
    float* d_head = nullptr;          // [max_batch][4+nc][N]
    Cand* d_cand = nullptr;           // [max_batch][N]
    Cand* d_scratch = nullptr;        // [max_batch][N]  (NMS spill when a frame has > 1024 candidates)
    int* d_count = nullptr;           // [max_batch]
    // ZLY_FLAG_ASYNC_NMS: second candidate buffer + a stream of its own for NMS, so that NMS of call k (64 workgroups,
    // latency-bound) runs beside the first kernels of call k+1 instead of idling the chip at the end of every step
    Cand* d_cand_alt = nullptr;
    int* d_count_alt = nullptr;
    Cand* cur_cand = nullptr;         // buffers the Detect tail / NMS of the current call use
    int* cur_count = nullptr;
    hipStream_t nms_stream = nullptr;
    hipEvent_t ev_head = nullptr, ev_nms[2] = {nullptr, nullptr};
    int parity = 0;
    bool nms_recorded[2] = {false, false};
    int last_async = -1;              // parity of the last deferred NMS, -1 = none outstanding
    unsigned char* d_slabs = nullptr; // [max_batch][slab_bytes]
    FrameDesc* d_desc = nullptr;      // [max_batch]
    // per-call descriptor ring (pinned): a call whose frame sizes differ from the previous call's writes the next ring entry and
    // uploads it in stream order; an entry is rewritten only after the upload that read it has completed (its event), so no
    // call ever waits for the device -- a mixed-client stream (800x600 next to 416x416) used to cost a hipDeviceSynchronize
    static constexpr int DESC_RING = 8;
    FrameDesc* h_desc = nullptr;      // pinned [DESC_RING][max_batch]
    hipEvent_t ev_desc[DESC_RING] = {};
    bool desc_used[DESC_RING] = {};
    int desc_next = 0;
    std::vector<FrameDesc> desc_cache;
    uint8_t* d_stage = nullptr;       // frame staging (host path)
    uint8_t* h_stage = nullptr;       // pinned
    size_t stage_bytes = 0;
    unsigned char* h_slabs = nullptr; // pinned
    float* d_scratch_f32 = nullptr;   // stage-level entry points
    size_t scratch_f32_elems = 0;

    bool stem_fused = false;          // bf16 + 16-channel stem: preprocess and model.0 are one kernel on the detect paths
    StemArgs stem{};
    bool stem1 = false;               // ... and model.1 (32 channels) as well: the stem map stays in LDS (kernels_stem.hip: stem_model1_kernel)
    Stem1Args stem1a{};
    bool ingest_active = false;       // set while the pipelined host path (zly_submit) enqueues: see run_path
    bool last_stem1 = false;          // the most recent call ran it (model.0 then only exists in HBM with ZLY_FLAG_DUMP_LOGITS)
    hipStream_t stream = nullptr;
    hipStream_t side[2] = {nullptr, nullptr};     // P3 / P4 Detect branches (forked from and joined to the main stream)
    hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join[2] = {nullptr, nullptr};
    std::map<int, hipGraphExec_t> graphs;   // (batch size, fused front, candidate-buffer parity) -> captured forward+decode
    std::map<int, int> graph_failures;      // failed captures per key (a second failure leaves the shape on eager launches)
    std::map<long, std::pair<bool, C2fPlan>> c2f_plans;     // (c, mode, n, H, W) -> fused C2f tile plan
    std::map<long, std::pair<bool, PairPlan>> pair_plans;   // (c, n, H, W) -> fused bottleneck tile plan (or "run unfused")
    int last_n = 0;

    // production phase timing: every SAMPLE_EVERY-th call of a detect path is bracketed by four timing events
    static constexpr int SAMPLE_EVERY = 16;
    hipEvent_t ev_t[4] = {nullptr, nullptr, nullptr, nullptr};
    bool t_pending = false;
    int t_frames = 0;
    unsigned sample_ctr = 0;

    // completion of the last two zly_detect_device calls (recorded behind each call's NMS, on whichever stream it ran): zly_join orders
    // a foreign stream behind them -- the RCCL gather of a step's slabs when the step ran on the engine's own stream (several engines
    // per GPU) or with deferred NMS
    hipEvent_t ev_call[2] = {nullptr, nullptr};
    uint64_t call_seq = 0;
    int det_stem[3] = {-1, -1, -1}, det_a[3] = {-1, -1, -1}, det_b[3] = {-1, -1, -1};   // op indices of the Detect convs (stem, box .1, class .1) per level
    int sppf_cv1 = -1, sppf_pool = -1, sppf_cv2 = -1;     // op indices of the SPPF block (model.9): one launch where kernels_sppf.hip covers the shape
    int cu_part_n = 1;                // ZLY_CU_PART: number of CU partitions (1 = whole chip)
    uint32_t cu_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::atomic<Ingest*> ingest{nullptr};   // created by the first zly_submit (under mu), read lock-free afterwards

    // tuning / test switches of the environment, read ONCE at zly_create (they used to be read per launch)
    struct Switches { bool no_c2f = false, no_det_merge = false, no_tail_split = false, no_lanes = false, nms_general = false, no_sppf = false, no_wsk = false, pool_six_pass = false; int stem1_nw = 0, stem1_var = 1, stem1_grid = 0; std::string ablate; } sw;

    std::mutex mu;                    // serialises every call that touches engine / device state
    mutable std::mutex stats_mu;      // guards `stats` only, never held across a device call: zly_get_stats cannot wait on a batch
    zly_stats stats{};
};

namespace zly {

static size_t slab_bytes_of(const zly_engine* e) { return sizeof(zly_slab_header) + (size_t)e->cfg.max_dets * sizeof(zly_det); }

template <typename F> static void with_stats(zly_engine* e, F&& f) { std::lock_guard<std::mutex> lk(e->stats_mu); f(e->stats); }

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
struct PlanBuilder {
    zly_engine* e;
    std::vector<uint8_t> blob;         // host image of the device weight blob
    std::string err;
    int kstep;

    int add_buffer(const std::string& name, int H, int W, int C, bool f32 = false) {
        Buffer b;
        b.name = name; b.H = H; b.W = W; b.C = C; b.f32 = f32;
        e->bufs.push_back(b);
        return (int)e->bufs.size() - 1;
    }
    size_t append(const void* p, size_t bytes) {
        size_t off = (blob.size() + 255) / 256 * 256;
        blob.resize(off + bytes);
        memcpy(blob.data() + off, p, bytes);
        return off;
    }
    bool conv(const std::vector<std::string>& names, View in, View out, View res = View{-1, 0, 0}, bool out_f32 = false, View in2 = View{-1, 0, 0}) {
        std::vector<const ConvRec*> srcs;
        for (const std::string& n : names) {
            const ConvRec* r = e->model.find(n);
            if (!r) { err = "conv missing from model file: " + n; return false; }
            srcs.push_back(r);
        }
        const ConvRec* r0 = srcs[0];
        for (const ConvRec* r : srcs)
            if (r->cin != r0->cin || r->k != r0->k || r->stride != r0->stride || r->act != r0->act) { err = "cannot fuse " + names[0]; return false; }
        const Buffer& ib = e->bufs[(size_t)in.buf];
        const Buffer& ob = e->bufs[(size_t)out.buf];
        const int epl = e->dtype == ZLY_DTYPE_BF16 ? 8 : 4;
        const int cin_total = in.C + (in2.buf >= 0 ? in2.C : 0);
        if (cin_total < r0->cin || in.C % epl != 0 || in.co % epl != 0 || ib.C % epl != 0 || in.co + in.C > ib.C) { err = "bad input view for " + names[0]; return false; }
        if (in2.buf >= 0) {
            const Buffer& i2 = e->bufs[(size_t)in2.buf];
            if (r0->k != 1 || in.C % kstep != 0 || in2.C % epl != 0 || in2.co % epl != 0 || i2.C % epl != 0 || in2.co + in2.C > i2.C ||
                i2.H != ob.H || i2.W != ob.W || ib.H * 2 != ob.H || ib.W * 2 != ob.W) { err = "bad upsample+concat input for " + names[0]; return false; }
        }
        Op op;
        op.kind = OP_CONV;
        op.name = names[0];
        for (size_t i = 1; i < names.size(); ++i) op.name += "+" + names[i];
        if (op.name.size() > 47) op.name.resize(47);
        op.in = in; op.out = out; op.res = res; op.in2 = in2;
        op.ks = r0->k; op.stride = r0->stride; op.act = r0->act; op.out_f32 = out_f32 ? 1 : 0;
        std::vector<uint8_t> w;
        std::vector<float> b;
        repack_conv(srcs, cin_total, kstep, e->dtype == ZLY_DTYPE_BF16, &w, &b, &op.cout, &op.cout_pad, &op.nk, true);
        // 5 channel tiles (the 80-channel Detect class branch) suit no kernel's channel blocking: CT = 5 needs more registers than
        // two waves per SIMD leave.  One whole zero tile is appended instead (6 tiles = 2 blocks of CT = 3); its outputs are never
        // stored (channel >= Cout) -- 20 % more MFMAs on these layers, but the LDS-tiled kernel instead of the direct one.
        if (e->dtype == ZLY_DTYPE_BF16 && op.ks == 3 && op.cout_pad == 80 && cin_total % 32 == 0 && getenv("ZLY_NO_COUT_PAD") == nullptr) {
            w.resize(w.size() + (size_t)op.nk * 1024, 0);
            b.resize(b.size() + 16, 0.0f);
            op.cout_pad = 96;
        }
        if (out.C != op.cout || out.co % 4 != 0 || ob.C % 4 != 0 || out.co + out.C > ob.C) { err = "bad output view for " + names[0]; return false; }
        if (out_f32 != ob.f32) { err = "output dtype mismatch for " + names[0]; return false; }
        const int Ho = ob.H, Wo = ob.W;
        const int Hin = in2.buf >= 0 ? ib.H * 2 : ib.H, Win = in2.buf >= 0 ? ib.W * 2 : ib.W;
        if (Ho != (Hin + 2 * (op.ks / 2) - op.ks) / op.stride + 1 || Wo != (Win + 2 * (op.ks / 2) - op.ks) / op.stride + 1) { err = "spatial mismatch for " + names[0]; return false; }
        op.K = op.ks * op.ks * cin_total;
        op.w_off = append(w.data(), w.size());
        op.b_off = append(b.data(), b.size() * sizeof(float));
        int co = 0;
        double macs = 0;
        for (size_t i = 0; i < srcs.size(); ++i) {
            op.taps.push_back(names[i]); op.tap_co.push_back(co); op.tap_c.push_back(srcs[i]->cout);
            co += srcs[i]->cout;
            macs += (double)Ho * Wo * srcs[i]->cout * srcs[i]->cin * op.ks * op.ks;
        }
        op.flops = 2.0 * macs;
        const double osz = out_f32 ? 4.0 : (double)e->esz;
        op.cin_frac = (double)r0->cin / cin_total;
        op.wbytes = (double)op.cout * r0->cin * op.ks * op.ks * e->esz + (double)op.cout * 4.0;
        op.bytes = ((double)ib.H * ib.W * in.C + (in2.buf >= 0 ? (double)Ho * Wo * in2.C : 0.0)) * e->esz * op.cin_frac + (double)Ho * Wo * op.cout * osz +
                   op.wbytes + (res.buf >= 0 ? (double)Ho * Wo * op.cout * e->esz : 0.0);
        e->ops.push_back(op);
        for (size_t i = 0; i < names.size(); ++i) e->tap_index[names[i]] = std::make_pair((int)e->ops.size() - 1, (int)i);
        return true;
    }
    // repack + upload the weights of a conv that is executed inside another kernel (fused Detect tail)
    bool pack_only(const std::string& name, int cin_store, size_t* w_off, size_t* b_off, int* nk, bool pair_rows = false) {
        const ConvRec* r = e->model.find(name);
        if (!r) { err = "conv missing from model file: " + name; return false; }
        std::vector<uint8_t> w;
        std::vector<float> b;
        int cout = 0, cout_pad = 0;
        repack_conv({r}, cin_store, kstep, e->dtype == ZLY_DTYPE_BF16, &w, &b, &cout, &cout_pad, nk, pair_rows);
        *w_off = append(w.data(), w.size());
        *b_off = append(b.data(), b.size() * sizeof(float));
        return true;
    }
    // The two convs just added are a bottleneck's 3x3 pair: when the fused kernel covers this width, the first op
    // also carries the pair kernel's operands (for c = 16 its own weight tiling: one tap per 16x16x16 MFMA k-step).
    bool mark_pair(const std::string& m, int c, bool shortcut) {
        if (e->dtype != ZLY_DTYPE_BF16 || (c != 16 && c != 32 && c != 64) || e->ops.size() < 2) return true;
        Op& B = e->ops[e->ops.size() - 1];
        Op& A = e->ops[e->ops.size() - 2];
        if (A.ks != 3 || B.ks != 3 || A.stride != 1 || B.stride != 1 || !A.act || !B.act || A.cout != c || B.cout != c) return true;
        if (A.in.co % 8 || B.out.co % 8 || e->bufs[(size_t)A.in.buf].C % 8 || e->bufs[(size_t)B.out.buf].C % 8) return true;
        A.pair = 1; B.pair = 2; A.pair_c = B.pair_c = c; A.pair_res = shortcut ? 1 : 0;
        A.pair_out = B.out;
        if (c >= 32) { A.pair_wA = A.w_off; A.pair_bA = A.b_off; A.pair_wB = B.w_off; A.pair_bB = B.b_off; return true; }
        const char* names[2] = {".cv1", ".cv2"};
        size_t* wo[2] = {&A.pair_wA, &A.pair_wB};
        size_t* bo[2] = {&A.pair_bA, &A.pair_bB};
        for (int k = 0; k < 2; ++k) {
            const ConvRec* r = e->model.find(m + names[k]);
            if (!r) { err = "conv missing from model file: " + m + names[k]; return false; }
            std::vector<uint8_t> w;
            std::vector<float> b;
            int cout = 0, cout_pad = 0, nk = 0;
            repack_conv({r}, c, 16, true, &w, &b, &cout, &cout_pad, &nk, false, 4);
            if (nk != 9 || cout_pad != 16) { err = "internal: pair weight tiling for " + m; return false; }
            *wo[k] = append(w.data(), w.size());
            *bo[k] = append(b.data(), b.size() * sizeof(float));
        }
        return true;
    }
    // C2f(c1 -> c2, n bottlenecks): cv1 writes [0,2c) of the concat buffer, bottleneck i reads
    // [(1+i)c,(2+i)c) and writes [(2+i)c,(3+i)c), cv2 reads all (2+n)c channels.
    bool c2f(const std::string& p, View in, View out, int n, bool shortcut, int H, int W, View in2 = View{-1, 0, 0}) {
        const int c = out.C / 2;
        const int cat = add_buffer(p + ".cat", H, W, (2 + n) * c);
        if (!conv({p + ".cv1"}, in, View{cat, 0, 2 * c}, View{-1, 0, 0}, false, in2)) return false;
        for (int i = 0; i < n; ++i) {
            const View src{cat, (1 + i) * c, c};
            const std::string m = p + ".m." + std::to_string(i);
            const int tmp = add_buffer(m + ".tmp", H, W, c);
            if (!conv({m + ".cv1"}, src, View{tmp, 0, c})) return false;
            if (!conv({m + ".cv2"}, View{tmp, 0, c}, View{cat, (2 + i) * c, c}, shortcut ? src : View{-1, 0, 0})) return false;
            if (!mark_pair(m, c, shortcut)) return false;
        }
        if (!conv({p + ".cv2"}, View{cat, 0, (2 + n) * c}, out)) return false;
        return mark_c2f(p, cat, c, n, shortcut, in, in2, out);
    }
    // The 2 + 2n ops just added are a C2f block: annotate the fused kernel's launch groups (n = 1: the whole block; n = 2: front half
    // = cv1 + first bottleneck, back half = second bottleneck + cv2).  Weight tilings are shared with the per-conv kernels where the
    // layout is the same (C = 32); the 16-channel block gets its own cv1 (rows in channel order) and cv2 (k-steps of 16) tilings.
    bool mark_c2f(const std::string& p, int cat, int c, int n, bool shortcut, View in, View in2, View out) {
        if (e->dtype != ZLY_DTYPE_BF16 || (c != 16 && c != 32 && c != 64) || (n != 1 && n != 2) || (e->cfg.flags & ZLY_FLAG_NO_FUSION)) return true;
        const size_t base = e->ops.size() - (size_t)(2 + 2 * n);
        Op& cv1 = e->ops[base];
        Op& cv2 = e->ops[e->ops.size() - 1];
        const int cin = in.C + (in2.buf >= 0 ? in2.C : 0);
        if (cin % 32 != 0 || cv2.cout != 2 * c || cv1.cout != 2 * c || !cv1.act || !cv2.act) return true;
        for (int i = 0; i < n; ++i) if (e->ops[base + 1 + 2 * (size_t)i].pair != 1) return true;       // mark_pair declined
        size_t w1 = cv1.w_off, b1 = cv1.b_off, w2 = cv2.w_off, b2 = cv2.b_off;
        int nk1 = cv1.nk, nk2 = cv2.nk;
        if (c == 16) {
            const ConvRec* r1 = e->model.find(p + ".cv1");
            const ConvRec* r2 = e->model.find(p + ".cv2");
            std::vector<uint8_t> w;
            std::vector<float> b;
            int cout = 0, cout_pad = 0;
            repack_conv({r1}, cin, 32, true, &w, &b, &cout, &cout_pad, &nk1, false);
            w1 = append(w.data(), w.size()); b1 = append(b.data(), b.size() * sizeof(float));
            repack_conv({r2}, (2 + n) * c, 16, true, &w, &b, &cout, &cout_pad, &nk2, true, 4);
            w2 = append(w.data(), w.size()); b2 = append(b.data(), b.size() * sizeof(float));
        }
        if (nk2 != (c == 64 ? 2 : 1) * (2 + n)) return true;            // cv2's k-steps: one per source map (c = 16 / 32), two per map of 64 channels
        auto fill = [&](Op& L, int mode, const Op& A) {
            L.c2f_mode = mode; L.c2f_c = c; L.c2f_res = shortcut ? 1 : 0; L.c2f_cat = cat;
            L.c2f_in_co = A.in.co; L.c2f_out_co = A.pair_out.co;
            L.c2f_x = in; L.c2f_x2 = in2; L.c2f_out = out;
            L.c2f_w1 = w1; L.c2f_b1 = b1; L.c2f_nk1 = nk1; L.c2f_w2 = w2; L.c2f_b2 = b2; L.c2f_nk2 = nk2; L.c2f_cout2 = cv2.cout;
            L.c2f_wA = A.pair_wA; L.c2f_bA = A.pair_bA; L.c2f_wB = A.pair_wB; L.c2f_bB = A.pair_bB;
            L.c2f_leader_name = L.name;
            if (c == 64) L.c2f_mid = A.out;                      // the 64-channel kernel can dump the bottleneck's intermediate map (debug taps)
        };
        if (n == 1) {
            Op& A = e->ops[base + 1];
            fill(cv1, 3, A);
            cv1.c2f_vis = 1;
            A.c2f_leader_name = cv1.name; A.c2f_vis = c == 64 ? 1 : 2;
            e->ops[base + 2].c2f_leader_name = cv1.name; e->ops[base + 2].c2f_vis = 1;
            cv2.c2f_leader_name = cv1.name; cv2.c2f_vis = 0;
        } else {
            Op& A0 = e->ops[base + 1];
            Op& A1 = e->ops[base + 3];
            fill(cv1, 1, A0);
            cv1.c2f_vis = 0;
            A0.c2f_leader_name = cv1.name; A0.c2f_vis = c == 64 ? 1 : 2;
            e->ops[base + 2].c2f_leader_name = cv1.name; e->ops[base + 2].c2f_vis = 0;
            fill(A1, 2, A1);
            A1.c2f_vis = c == 64 ? 1 : 2;
            e->ops[base + 4].c2f_leader_name = A1.name; e->ops[base + 4].c2f_vis = 1;
            cv2.c2f_leader_name = A1.name; cv2.c2f_vis = 0;
        }
        return true;
    }
};

// build_plan runs in two halves: the HOST half -- the op list and every weight repack into a host image of the device blob -- needs no
// device call and runs before zly_create takes the process-wide exclusive gate; the DEVICE half (hipMalloc / hipMemcpy, pointers into the blob) runs
// under it.  A hot reload builds its new engines beside running ones: their enqueue sections now stall for the allocations only, not for the repack.
struct PendingLevel { size_t wb, bb, wc, bc; int hb2, hc2, hout; };
struct PlanState {
    std::vector<uint8_t> blob;
    size_t stem_w = 0, stem_b = 0, m1_w = 0, m1_b = 0, stem_w0p = 0;
    int a0 = -1, a1 = -1;
    PendingLevel pend[3] = {};
};

static int build_plan_host(zly_engine* e, PlanState* ps, std::string* err)
{
    PlanBuilder pb;
    pb.e = e;
    pb.kstep = conv_kstep(e->dtype);
    const ModelFile& m = e->model;
    const int W = e->cfg.model_w, H = e->cfg.model_h;
    const int* ch = m.ch;
    const int* nb = m.n_c2f;
    if (m.reg_max != 16) { *err = "only reg_max = 16 is supported"; return ZLY_ERR_MODEL_LOAD; }
    for (int i = 0; i < 5; ++i)
        if (ch[i] % 16 != 0) { *err = "channel widths must be multiples of 16"; return ZLY_ERR_MODEL_LOAD; }
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8, H16 = H / 16, W16 = W / 16, H32 = H / 32, W32 = W / 32;

    e->in_buf = pb.add_buffer("images", H, W, 8);
    const int a0 = pb.add_buffer("model.0", H2, W2, ch[0]);
    const int a1 = pb.add_buffer("model.1", H4, W4, ch[1]);
    const int a2 = pb.add_buffer("model.2", H4, W4, ch[1]);
    const int a3 = pb.add_buffer("model.3", H8, W8, ch[2]);
    const int cat14 = pb.add_buffer("cat14[up(12),4]", H8, W8, ch[3] + ch[2]);
    const int a5 = pb.add_buffer("model.5", H16, W16, ch[3]);
    const int cat11 = pb.add_buffer("cat11[up(9),6]", H16, W16, ch[4] + ch[3]);
    const int a7 = pb.add_buffer("model.7", H32, W32, ch[4]);
    const int a8 = pb.add_buffer("model.8", H32, W32, ch[4]);
    const int spp = pb.add_buffer("model.9.cat", H32, W32, 2 * ch[4]);
    const int cat20 = pb.add_buffer("cat20[19,9]", H32, W32, ch[3] + ch[4]);
    const int cat17 = pb.add_buffer("cat17[16,12]", H16, W16, ch[2] + ch[3]);
    const int a15 = pb.add_buffer("model.15", H8, W8, ch[2]);
    const int a18 = pb.add_buffer("model.18", H16, W16, ch[3]);
    const int a21 = pb.add_buffer("model.21", H32, W32, ch[4]);

    Op pre; pre.kind = OP_PREPROCESS; pre.name = "preprocess";
    pre.bytes = (double)W * H * 3 + (double)W * H * 3 * e->esz;          // algorithmic: u8 frame in, the 3-channel tensor out (stored as 8 channels)
    e->ops.push_back(pre);

    bool ok = true;
    ok = ok && pb.conv({"model.0"}, View{e->in_buf, 0, 8}, View{a0, 0, ch[0]});
    size_t stem_w = 0, stem_b = 0;
    int stem_nk = 0;
    e->stem_fused = e->dtype == ZLY_DTYPE_BF16 && (ch[0] == 16 || ch[0] == 32);          // YOLOv8n / YOLOv8-s stems: preprocess + model.0 in one kernel
    if (e->stem_fused) ok = ok && pb.pack_only("model.0", 4, &stem_w, &stem_b, &stem_nk, ch[0] == 32) && stem_nk == 2;
    ok = ok && pb.conv({"model.1"}, View{a0, 0, ch[0]}, View{a1, 0, ch[1]});
    size_t m1_w = 0, m1_b = 0, stem_w0p = 0;
    e->stem1 = e->stem_fused && ch[0] == 16 && ch[1] == 32 && !(e->cfg.flags & ZLY_FLAG_NO_FUSION) && getenv("ZLY_NO_STEM1") == nullptr && ok;
    if (e->stem1) {
        // model.1 in the fused kernel's tiling: one 16x16x16 MFMA per tap (k = ci), pair-permuted rows, bias in channel order
        const ConvRec* r = m.find("model.1");
        std::vector<uint8_t> w;
        std::vector<float> b;
        int cout = 0, cout_pad = 0, nk = 0;
        repack_conv({r}, ch[0], 16, true, &w, &b, &cout, &cout_pad, &nk, true, 4);
        if (nk != 9 || cout_pad != 32 || r->k != 3 || r->stride != 2) e->stem1 = false;
        else { m1_w = pb.append(w.data(), w.size()); m1_b = pb.append(b.data(), b.size() * sizeof(float)); }
        // the stem's weights once more in the tap order of the fused kernel's conflict-free fragment reads (kernels_stem.hip: STEM1_TAP_SLOT)
        if (e->stem1) {
            const ConvRec* r0 = m.find("model.0");
            int nk0 = 0;
            repack_conv({r0}, 4, pb.kstep, true, &w, &b, &cout, &cout_pad, &nk0, false, 0, stem1_tap_slot());
            if (nk0 != 2 || cout_pad != 16) e->stem1 = false;
            else stem_w0p = pb.append(w.data(), w.size());
        }
    }
    ok = ok && pb.c2f("model.2", View{a1, 0, ch[1]}, View{a2, 0, ch[1]}, nb[0], true, H4, W4);
    ok = ok && pb.conv({"model.3"}, View{a2, 0, ch[1]}, View{a3, 0, ch[2]});
    ok = ok && pb.c2f("model.4", View{a3, 0, ch[2]}, View{cat14, ch[3], ch[2]}, nb[1], true, H8, W8);          // P3 -> cat14
    ok = ok && pb.conv({"model.5"}, View{cat14, ch[3], ch[2]}, View{a5, 0, ch[3]});
    ok = ok && pb.c2f("model.6", View{a5, 0, ch[3]}, View{cat11, ch[4], ch[3]}, nb[2], true, H16, W16);        // P4 -> cat11
    ok = ok && pb.conv({"model.7"}, View{cat11, ch[4], ch[3]}, View{a7, 0, ch[4]});
    ok = ok && pb.c2f("model.8", View{a7, 0, ch[4]}, View{a8, 0, ch[4]}, nb[3], true, H32, W32);
    // SPPF
    ok = ok && pb.conv({"model.9.cv1"}, View{a8, 0, ch[4]}, View{spp, 0, ch[4] / 2});
    if (ok) {
        Op p; p.kind = OP_SPPF; p.name = "model.9.pool"; p.in = View{spp, 0, ch[4] / 2}; p.c = ch[4] / 2;
        p.bytes = (double)H32 * W32 * (ch[4] / 2) * 4 * e->esz;
        e->ops.push_back(p);
    }
    ok = ok && pb.conv({"model.9.cv2"}, View{spp, 0, 2 * ch[4]}, View{cat20, ch[3], ch[4]});                     // P5 -> cat20
    // neck, top-down.  Upsample(x2, nearest) + Concat are fused into the consumer: model.12.cv1 / model.15.cv1 read
    // their first channels from the half-size tensor at (y>>1, x>>1) and the rest from the skip tensor.
    ok = ok && pb.c2f("model.12", View{cat20, ch[3], ch[4]}, View{cat17, ch[2], ch[3]}, nb[4], false, H16, W16, View{cat11, ch[4], ch[3]});
    ok = ok && pb.c2f("model.15", View{cat17, ch[2], ch[3]}, View{a15, 0, ch[2]}, nb[5], false, H8, W8, View{cat14, ch[3], ch[2]});
    // neck, bottom-up
    ok = ok && pb.conv({"model.16"}, View{a15, 0, ch[2]}, View{cat17, 0, ch[2]});
    ok = ok && pb.c2f("model.18", View{cat17, 0, ch[2] + ch[3]}, View{a18, 0, ch[3]}, nb[6], false, H16, W16);
    ok = ok && pb.conv({"model.19"}, View{a18, 0, ch[3]}, View{cat20, 0, ch[3]});
    ok = ok && pb.c2f("model.21", View{cat20, 0, ch[3] + ch[4]}, View{a21, 0, ch[4]}, nb[7], false, H32, W32);
    // Detect: per level, the two branches' first 3x3 convs read the same tensor and are one launch
    const int c2 = std::max(16, std::max(ch[2] / 4, 4 * m.reg_max));
    const int c3 = std::max(ch[2], std::min(m.nc, 100));
    const int feats[3] = {a15, a18, a21};
    const int fch[3] = {ch[2], ch[3], ch[4]};
    const int fh[3] = {H8, H16, H32}, fw[3] = {W8, W16, W32};
    const int ncp = (m.nc + 3) / 4 * 4;
    int anchor_off = 0;
    e->N = fh[0] * fw[0] + fh[1] * fw[1] + fh[2] * fw[2];
    if (m.nc > 80) { *err = "nc > 80 is not supported by the fused Detect kernel"; return ZLY_ERR_MODEL_LOAD; }
    Op hd;
    hd.kind = OP_HEAD; hd.name = "detect.tail(1x1 convs+DFL+sigmoid+decode)";
    if (hd.name.size() > 47) hd.name.resize(47);
    PendingLevel (&pend)[3] = ps->pend;
    int block0 = 0;
    for (int l = 0; l < 3 && ok; ++l) {
        const std::string L = std::to_string(l);
        // the class branch's 80 channels are stored as 96 (16 zero channels, never written) so that its second 3x3 conv has
        // Cin % 32 == 0 and takes the LDS-tiled kernel; the padded k-steps carry zero weights
        const int c3s = (e->dtype == ZLY_DTYPE_BF16 && c3 % 32 != 0 && c2 % 32 == 0 && getenv("ZLY_NO_CIN_PAD") == nullptr) ? (c3 + 31) / 32 * 32 : c3;
        const int hd1 = pb.add_buffer("detect." + L + ".stem", fh[l], fw[l], c2 + c3s);
        const int hb2 = pb.add_buffer("detect." + L + ".box2", fh[l], fw[l], c2);
        const int hc2 = pb.add_buffer("detect." + L + ".cls2", fh[l], fw[l], c3);
        const int hout = pb.add_buffer("detect." + L + ".logits", fh[l], fw[l], 64 + ncp, true);
        ok = ok && pb.conv({"model.22.cv2." + L + ".0", "model.22.cv3." + L + ".0"}, View{feats[l], 0, fch[l]}, View{hd1, 0, c2 + c3});
        ok = ok && pb.conv({"model.22.cv2." + L + ".1"}, View{hd1, 0, c2}, View{hb2, 0, c2});
        ok = ok && pb.conv({"model.22.cv3." + L + ".1"}, View{hd1, c2, c3s}, View{hc2, 0, c3});
        if (ok && e->dtype == ZLY_DTYPE_BF16 && c3 == 80 && c3s >= 80) {
            // the same conv for conv3x3_wsk_kernel: k = tap * 80 + c without channel padding (23 k-steps instead of 27), plain tile rows, no zero output tile
            const ConvRec* r = m.find("model.22.cv3." + L + ".1");
            std::vector<uint8_t> w;
            std::vector<float> b;
            int cout = 0, cout_pad = 0, nk = 0;
            repack_conv({r}, 80, pb.kstep, true, &w, &b, &cout, &cout_pad, &nk, false);
            if (nk == 23 && cout_pad == 80 && r->k == 3 && r->stride == 1) e->ops.back().wsk_w = pb.append(w.data(), w.size());
        }
        HeadLevel& hl = hd.head.lv[l];
        ok = ok && pb.pack_only("model.22.cv2." + L + ".2", c2, &pend[l].wb, &pend[l].bb, &hl.nkb);
        ok = ok && pb.pack_only("model.22.cv3." + L + ".2", c3, &pend[l].wc, &pend[l].bc, &hl.nkc);
        pend[l].hb2 = hb2; pend[l].hc2 = hc2; pend[l].hout = hout;
        hl.box_cs = c2; hl.cls_cs = c3; hl.box_cin = c2; hl.cls_cin = c3;
        hl.H = fh[l]; hl.W = fw[l]; hl.hw = fh[l] * fw[l]; hl.stride_px = 8 << l; hl.anchor_off = anchor_off; hl.block0 = block0;
        hl.logits_cs = 64 + ncp;
        block0 += (hl.hw + HEAD_GROUP - 1) / HEAD_GROUP;
        const double macs = (double)hl.hw * (c2 * 64.0 + (double)c3 * m.nc);
        hd.flops += 2.0 * macs;
        // algorithmic bytes: both branch activations in, the weights, and -- only when the engine materialises it -- the fp32 head tensor out
        hd.bytes += (double)hl.hw * ((c2 + c3) * (double)e->esz + ((e->cfg.flags & ZLY_FLAG_NO_HEAD_TENSOR) ? 0.0 : (4 + m.nc) * 4.0)) + (c2 * 64.0 + (double)c3 * m.nc) * e->esz;
        hd.wbytes += (c2 * 64.0 + (double)c3 * m.nc) * e->esz;
        hd.head_box[l] = hb2; hd.head_cls[l] = hc2;
        e->tap_final[std::string("model.22.cv2.") + L + ".2"] = std::make_pair(hout, 0);
        e->tap_final[std::string("model.22.cv3.") + L + ".2"] = std::make_pair(hout, 64);
        e->lvl_h[l] = fh[l]; e->lvl_w[l] = fw[l];
        anchor_off += fh[l] * fw[l];
    }
    hd.head.nc = m.nc; hd.head.N_total = e->N; hd.head.total_blocks = block0; hd.head.only_level = -1;
    // Three tail ops, one per level.  From batch 16 up each is its own launch right behind its level's branch convs (P3 and
    // P4 on the side streams, i.e. beside the neck) and only the small P5 launch is left at the end of the step; below
    // that the first two are skipped and the last one covers all levels in a single launch (fewest launches).
    for (int l = 0; l < 3 && ok; ++l) {
        Op t = hd;
        t.level = l;
        t.name = std::string("detect.tail.P") + std::to_string(3 + l) + (l == 2 ? " (all levels below batch 16)" : "");
        const double share = (double)hd.head.lv[l].hw / (double)e->N;
        t.flops = hd.flops * share; t.bytes = hd.bytes * share; t.wbytes = hd.wbytes * share;
        e->ops.push_back(t);
    }
    if (!ok) { *err = pb.err; return ZLY_ERR_MODEL_LOAD; }
    Op nm; nm.kind = OP_NMS; nm.name = "nms"; e->ops.push_back(nm);

    // The Detect branches of P3 and P4 only depend on model.15 / model.18: issue them right behind their
    // producer on side streams (lanes 1, 2) so that, inside the captured graph, they run BESIDE the rest of
    // the neck (model.16-21) instead of after it.  These launches are latency/issue-bound and leave most of
    // the chip idle, so the overlap is nearly free.  The tail op joins all lanes.
    {
        std::vector<Op> main_ops, lane_ops[3];
        for (Op& op : e->ops) {
            int lane = 0;
            if (op.kind == OP_CONV && op.name.rfind("model.22.cv", 0) == 0) {
                const int lvl = op.name[13] - '0';                 // "model.22.cvX.L...": L is character 13
                lane = lvl == 0 ? 1 : (lvl == 1 ? 2 : 0);
            }
            if (op.kind == OP_HEAD) lane = op.level == 0 ? 1 : (op.level == 1 ? 2 : 0);
            op.lane = lane;
            (lane == 0 ? main_ops : lane_ops[lane]).push_back(op);
        }
        std::vector<Op> ordered;
        for (Op& op : main_ops) {
            ordered.push_back(op);
            if (op.kind == OP_CONV && op.name == "model.15.cv2") for (Op& x : lane_ops[1]) ordered.push_back(x);
            if (op.kind == OP_CONV && op.name == "model.18.cv2") for (Op& x : lane_ops[2]) ordered.push_back(x);
        }
        if (ordered.size() != e->ops.size()) { *err = "internal: op reordering lost ops"; return ZLY_ERR_MODEL_LOAD; }
        e->ops.swap(ordered);
        e->tap_index.clear();
        for (size_t i = 0; i < e->ops.size(); ++i)
            for (size_t k = 0; k < e->ops[i].taps.size(); ++k) e->tap_index[e->ops[i].taps[k]] = std::make_pair((int)i, (int)k);
        std::map<std::string, int> by_name;
        for (size_t i = 0; i < e->ops.size(); ++i) by_name[e->ops[i].name] = (int)i;
        for (Op& op : e->ops)
            if (!op.c2f_leader_name.empty()) op.c2f_leader = by_name[op.c2f_leader_name];
        if (by_name.count("model.9.cv1") && by_name.count("model.9.pool") && by_name.count("model.9.cv2")) {
            e->sppf_cv1 = by_name["model.9.cv1"]; e->sppf_pool = by_name["model.9.pool"]; e->sppf_cv2 = by_name["model.9.cv2"];
        }
        // Detect convs, for the merged launches of the latency path: all three levels must suit the shared kernel shape
        bool det_ok = e->dtype == ZLY_DTYPE_BF16;
        for (int l = 0; l < 3 && det_ok; ++l) {
            const std::string L = std::to_string(l);
            for (size_t i = 0; i < e->ops.size(); ++i) {
                const Op& op = e->ops[i];
                if (op.kind != OP_CONV) continue;
                if (op.name.rfind("model.22.cv2." + L + ".0", 0) == 0) e->det_stem[l] = (int)i;
                if (op.name == "model.22.cv2." + L + ".1") e->det_a[l] = (int)i;
                if (op.name == "model.22.cv3." + L + ".1") e->det_b[l] = (int)i;
            }
            det_ok = e->det_stem[l] >= 0 && e->det_a[l] >= 0 && e->det_b[l] >= 0;
            if (det_ok) {
                const Op& S = e->ops[(size_t)e->det_stem[l]];
                const Op& A = e->ops[(size_t)e->det_a[l]];
                const Op& Bc = e->ops[(size_t)e->det_b[l]];
                det_ok = S.ks == 3 && A.ks == 3 && Bc.ks == 3 && S.stride == 1 && S.in.C % 32 == 0 && A.in.C % 32 == 0 && Bc.in.C % 32 == 0 &&
                         S.cout_pad % 48 == 0 && A.cout_pad % 32 == 0 && Bc.cout_pad % 32 == 0 && S.in2.buf < 0 && S.res.buf < 0 && A.res.buf < 0 && Bc.res.buf < 0;
            }
        }
        if (!det_ok) e->det_stem[0] = -1;
    }

    ps->blob.swap(pb.blob);
    ps->stem_w = stem_w; ps->stem_b = stem_b; ps->m1_w = m1_w; ps->m1_b = m1_b; ps->stem_w0p = stem_w0p; ps->a0 = a0; ps->a1 = a1;
    return ZLY_OK;
}

static int build_plan_device(zly_engine* e, PlanState* ps, std::string* err)
{
    const ModelFile& m = e->model;
    const int W = e->cfg.model_w, H = e->cfg.model_h;
    const int* ch = m.ch;
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    const size_t stem_w = ps->stem_w, stem_b = ps->stem_b, m1_w = ps->m1_w, m1_b = ps->m1_b, stem_w0p = ps->stem_w0p;
    const int a0 = ps->a0, a1 = ps->a1;
    const PendingLevel (&pend)[3] = ps->pend;
    struct { std::vector<uint8_t>& blob; } pb{ps->blob};
    // device allocations
    const int B = e->cfg.max_batch;
    for (Buffer& b : e->bufs) {
        if (b.elems_per_frame() * (size_t)B >= (1ull << 31)) { *err = "activation buffer " + b.name + " exceeds 2^31 elements: lower max_batch"; return ZLY_ERR_INVALID_ARGUMENT; }
        const size_t bytes = b.elems_per_frame() * (b.f32 ? 4 : e->esz) * (size_t)B;
        if (hipMalloc(&b.ptr, bytes) != hipSuccess) { *err = "hipMalloc failed for " + b.name; return ZLY_ERR_SYSTEM; }
        hipMemset(b.ptr, 0, bytes);
    }
    if (hipMalloc(&e->d_weights, pb.blob.size()) != hipSuccess) { *err = "hipMalloc failed for weights"; return ZLY_ERR_SYSTEM; }
    if (hipMemcpy(e->d_weights, pb.blob.data(), pb.blob.size(), hipMemcpyHostToDevice) != hipSuccess) { *err = "weight upload failed"; return ZLY_ERR_SYSTEM; }
    if (e->stem_fused) {
        StemArgs& st = e->stem;
        st.wgt = (const char*)e->d_weights + stem_w; st.bias = (const float*)((const char*)e->d_weights + stem_b);
        st.out = e->bufs[(size_t)a0].ptr; st.out_cs = ch[0]; st.out_co = 0;
        st.tw = W; st.th = H; st.Ho = H2; st.Wo = W2; st.Cout = ch[0]; st.tiles_x = stem_tiles_x(W2);
    }
    if (e->stem1) {
        Stem1Args& s1 = e->stem1a;
        s1.st = e->stem;
        s1.w1 = (const char*)e->d_weights + m1_w; s1.b1 = (const float*)((const char*)e->d_weights + m1_b);
        s1.out1 = e->bufs[(size_t)a1].ptr; s1.out1_cs = ch[1]; s1.out1_co = 0;
        s1.H1 = H4; s1.W1 = W4;
        stem1_plan(H4, W4, &s1.TH, &s1.TW);
        s1.tiles_x = (W4 + s1.TW - 1) / s1.TW; s1.tiles_y = (H4 + s1.TH - 1) / s1.TH;
        s1.dump = (e->cfg.flags & ZLY_FLAG_DUMP_LOGITS) ? 1 : 0;
        s1.wgt0p = (const char*)e->d_weights + stem_w0p;
        s1.nw = e->sw.stem1_nw; s1.var = e->sw.stem1_var; s1.pgrid = e->sw.stem1_grid;
        s1.small_tiles = (getenv("ZLY_STEM1_TW") || getenv("ZLY_STEM1_TH") || getenv("ZLY_STEM1_BIG_TILES")) ? 0 : 1;      // tuning aids given: keep the planned shape at every batch size
    }
    for (Op& op : e->ops) {
        if (op.kind != OP_HEAD) continue;
        for (int l = 0; l < 3; ++l) {
            HeadLevel& hl = op.head.lv[l];
            hl.box_in = e->bufs[(size_t)pend[l].hb2].ptr;
            hl.cls_in = e->bufs[(size_t)pend[l].hc2].ptr;
            hl.wb = (const char*)e->d_weights + pend[l].wb; hl.bb = (const float*)((const char*)e->d_weights + pend[l].bb);
            hl.wc = (const char*)e->d_weights + pend[l].wc; hl.bc = (const float*)((const char*)e->d_weights + pend[l].bc);
            hl.logits = (e->cfg.flags & ZLY_FLAG_DUMP_LOGITS) ? (float*)e->bufs[(size_t)pend[l].hout].ptr : nullptr;
        }
    }
    return ZLY_OK;
}

// ------------------------------------------------------------------------------------------------
// execution
// ------------------------------------------------------------------------------------------------
// Detect branches on side streams (and per-level tail launches): measured at batch 1 the cross-stream edges cost more
// than the overlap buys (0.27 -> 0.34 ms/frame); from batch 16 up the branches are long enough to pay
static bool lanes_active(const zly_engine* e, int n)
{
    if (e->cfg.flags & ZLY_FLAG_SINGLE_CHAIN) return false;
    return n >= 16 && !e->sw.no_lanes;                               // ZLY_NO_LANES / ZLY_CU_PART: tuning aid; a CU partition runs one chain
}

// Fused bottleneck pair for this op at batch n?  Only degenerate launches (a handful of tiles) stay on the per-conv
// kernels.  Same answer for both ops of a pair.
static const PairPlan* pair_active(zly_engine* e, const Op& op, int n)
{
    if (!op.pair || e->dtype != ZLY_DTYPE_BF16 || (e->cfg.flags & ZLY_FLAG_NO_FUSION)) return nullptr;
    const Buffer& b = e->bufs[(size_t)(op.pair == 1 ? op.in.buf : op.out.buf)];
    const long key = (((long)op.pair_c * 4096 + n) * 4096 + b.H) * 4096 + b.W;
    auto it = e->pair_plans.find(key);
    if (it == e->pair_plans.end()) {
        PairPlan pl{};
        const char* mt = getenv("ZLY_PAIR_MIN_TILES");                 // tuning / tests: force the fused kernel onto small launches
        const int min_tiles = mt ? atoi(mt) : 32;                       // batch 1 (91 / 234 tiles): one fused launch beats two per-conv launches, 4370 -> 4475 fps
        const char* pw = getenv("ZLY_PAIR_WIDTHS");                   // bit mask of fused widths (16 | 32 | 64), default 16 | 32
        const int widths = pw ? atoi(pw) : 48;                          // 64: built and tested, but no faster than two launches (below)
        const bool ok = (widths & op.pair_c) && pair_plan(op.pair_c, n, b.H, b.W, &pl) && pl.total_tiles >= min_tiles;
        it = e->pair_plans.emplace(key, std::make_pair(ok, pl)).first;
    }
    return it->second.first ? &it->second.second : nullptr;
}

// Fused C2f kernel for the group led by `L` at batch n?  (plan cached per leader and batch size)
static const C2fPlan* c2f_active(zly_engine* e, const Op& L, int n)
{
    if (!L.c2f_mode || e->dtype != ZLY_DTYPE_BF16 || (e->cfg.flags & ZLY_FLAG_NO_FUSION)) return nullptr;
    if (e->sw.no_c2f) return nullptr;                                       // ZLY_NO_C2F: tuning / tests
    const Buffer& b = e->bufs[(size_t)L.c2f_cat];
    const long key = ((((((long)L.c2f_c * 4 + L.c2f_mode) * 16 + L.c2f_nk1) * 16 + L.c2f_nk2) * 4096 + n) * 4096 + b.H) * 4096 + b.W;   // nk1 / nk2: the LDS layout of the 64-channel kernel depends on them
    auto it = e->c2f_plans.find(key);
    if (it == e->c2f_plans.end()) {
        C2fPlan pl{};
        const bool ok = c2f_plan(L.c2f_c, L.c2f_mode, L.c2f_nk1, L.c2f_nk2, L.c2f_cout2, n, b.H, b.W, &pl);
        it = e->c2f_plans.emplace(key, std::make_pair(ok, pl)).first;
    }
    return it->second.first ? &it->second.second : nullptr;
}

// is this op's work done by an active fused C2f kernel launched at another op?
static bool c2f_covered(zly_engine* e, const Op& op, int n)
{
    return op.c2f_leader >= 0 && !op.c2f_mode && c2f_active(e, e->ops[(size_t)op.c2f_leader], n) != nullptr;
}

// launch arguments of a conv op at batch n
static ConvArgs make_conv_args(zly_engine* e, const Op& op, int n)
{
    const Buffer& ib = e->bufs[(size_t)op.in.buf];
    const Buffer& ob = e->bufs[(size_t)op.out.buf];
    ConvArgs a;
    a.in = ib.ptr; a.in_cs = ib.C; a.in_co = op.in.co;
    a.H = ib.H; a.W = ib.W; a.Cin = op.in.C;
    a.wgt = (const char*)e->d_weights + op.w_off;
    a.bias = (const float*)((const char*)e->d_weights + op.b_off);
    a.out = ob.ptr; a.out_cs = ob.C; a.out_co = op.out.co;
    a.Ho = ob.H; a.Wo = ob.W; a.Cout = op.cout; a.cout_pad = op.cout_pad;
    if (op.res.buf >= 0) { const Buffer& rb = e->bufs[(size_t)op.res.buf]; a.res = rb.ptr; a.res_cs = rb.C; a.res_co = op.res.co; }
    else { a.res = nullptr; a.res_cs = 0; a.res_co = 0; }
    a.stride = op.stride; a.pad = op.ks / 2;
    a.K = op.K; a.nk = op.nk; a.M = n * ob.H * ob.W; a.act = op.act; a.out_f32 = op.out_f32;
    a.in2 = nullptr; a.in2_cs = 0; a.in2_co = 0; a.split_c = 0;
    a.inv_wo = 0.f; a.inv_ho = 0.f;                                   // filled where the split-K kernel is launched
    if (op.in2.buf >= 0) {
        const Buffer& i2 = e->bufs[(size_t)op.in2.buf];
        a.in2 = i2.ptr; a.in2_cs = i2.C; a.in2_co = op.in2.co; a.split_c = op.in.C;
        a.H = ob.H; a.W = ob.W; a.Cin = op.in.C + op.in2.C;        // logical (full-size, concatenated) input
    }
    return a;
}

// kernel shape of the per-conv path for op at batch n: picked once per (op, batch size) -- conv_pick_config reads ~10 environment
// switches, which the eager path (partial batches of the pipelined host path) used to do for every conv of every call
static const ConvLaunch& conv_launch_of(zly_engine* e, const Op& op, int n)
{
    Op& mop = const_cast<Op&>(op);
    auto it = mop.launch_cache.find(n);
    if (it == mop.launch_cache.end()) {
        const Buffer& ob = e->bufs[(size_t)op.out.buf];
        const int cin = op.in.C + (op.in2.buf >= 0 ? op.in2.C : 0);
        ConvLaunch c;
        conv_pick_config(e->dtype, op.ks, op.stride, cin, op.cout_pad, n, ob.H, ob.W, &c,
                         op.in2.buf < 0 && op.res.buf < 0 && op.act && !op.out_f32 && op.cout % 32 == 0,
                         op.in2.buf < 0 && !op.out_f32 && op.act,
                         op.in2.buf >= 0 && op.ks == 1 && op.res.buf < 0 && op.act && !op.out_f32 && op.cout % 32 == 0 && op.cout_pad == op.cout);
        it = mop.launch_cache.emplace(n, c).first;
    }
    return it->second;
}

// Detect convs merged into two launches on the latency path (conv_igemm_multi_kernel): batch <= 4, no side streams in use
static bool detect_merge_active(zly_engine* e, int n)
{
    if (e->dtype != ZLY_DTYPE_BF16 || e->det_stem[0] < 0 || n > 4 || lanes_active(e, n) || e->sw.no_det_merge) return false;
    return true;
}
// role of op index i in the merged Detect launches: 0 none, 1 covered (no launch), 2 launches the three stems, 3 launches the six branch convs
static int detect_merge_role(const zly_engine* e, int i)
{
    for (int l = 0; l < 3; ++l) {
        if (i == e->det_stem[l]) return l == 2 ? 2 : 1;
        if (i == e->det_a[l]) return l == 2 ? 3 : 1;
        if (i == e->det_b[l]) return 1;
    }
    return 0;
}

// the 80 -> 80 class-branch conv on the K-packed weight-stationary kernel at this batch size?  (decided once per op and batch size)
static bool wsk_active(zly_engine* e, const Op& op, int n)
{
    if (!op.wsk_w || e->sw.no_wsk || e->dtype != ZLY_DTYPE_BF16 || !op.act) return false;
    Op& mop = const_cast<Op&>(op);
    auto it = mop.wsk_cache.find(n);
    if (it == mop.wsk_cache.end()) {
        const Buffer& ob = e->bufs[(size_t)op.out.buf];
        it = mop.wsk_cache.emplace(n, conv_wsk_ok(80, op.cout, n, ob.H, ob.W) ? 1 : 0).first;
    }
    return it->second != 0;
}

// SPPF as one launch (kernels_sppf.hip) at the block's first op; the pool and cv2 ops then launch nothing
static bool sppf_active(const zly_engine* e)
{
    if (e->dtype != ZLY_DTYPE_BF16 || e->sppf_cv1 < 0 || (e->cfg.flags & ZLY_FLAG_NO_FUSION) || e->sw.no_sppf) return false;
    const Op& a = e->ops[(size_t)e->sppf_cv1];
    const Op& b = e->ops[(size_t)e->sppf_cv2];
    const Buffer& ob = e->bufs[(size_t)b.out.buf];
    if (!a.act || !b.act || a.res.buf >= 0 || b.res.buf >= 0 || a.in2.buf >= 0 || b.in.C != 4 * a.cout || a.cout_pad != a.cout || b.cout_pad != b.cout) return false;
    return sppf_fused_ok(a.in.C, a.cout, b.cout, ob.H, ob.W);
}

// ops that launch nothing at this batch size (second conv of a fused pair; per-level tail ops when one launch covers all)
static bool op_is_noop(zly_engine* e, const Op& op, int n)
{
    if (op.kind == OP_PREPROCESS) return e->stem_fused;                          // detect paths: inside the stem kernel
    if (op.kind == OP_CONV && e->stem1 && op.name == "model.1") return true;     // detect paths: computed by stem_model1_kernel (booked on model.0)
    if (op.kind == OP_CONV && c2f_covered(e, op, n)) return true;
    if ((op.kind == OP_SPPF || (op.kind == OP_CONV && (int)(&op - e->ops.data()) == e->sppf_cv2)) && sppf_active(e)) return true;
    if (op.kind == OP_CONV && detect_merge_active(e, n) && detect_merge_role(e, (int)(&op - e->ops.data())) == 1) return true;
    if (op.kind == OP_CONV && op.pair == 2) return pair_active(e, op, n) != nullptr;
    if (op.kind == OP_HEAD && op.level != 2) return !lanes_active(e, n) || e->sw.no_tail_split;
    return false;
}

// Launch groups (bookkeeping for zly_launch_info_at): index of the op whose launch does op i's work at batch n on the detect paths --
// i itself when it launches, the fused kernel's leader otherwise.
static int op_covered_by(zly_engine* e, int i, int n)
{
    const Op& op = e->ops[(size_t)i];
    if (op.kind == OP_PREPROCESS) return e->stem_fused ? 1 : i;
    if (op.kind == OP_CONV) {
        if (i == 2 && e->stem1) return 1;
        if (c2f_covered(e, op, n)) return op.c2f_leader;
        if (i == e->sppf_cv2 && sppf_active(e)) return e->sppf_cv1;
        if (detect_merge_active(e, n) && detect_merge_role(e, i) == 1) {
            for (int l = 0; l < 3; ++l) if (i == e->det_stem[l]) return e->det_stem[2];
            return e->det_a[2];
        }
        if (op.pair == 2 && pair_active(e, op, n)) return i - 1;
        return i;
    }
    if (op.kind == OP_SPPF && sppf_active(e)) return e->sppf_cv1;
    if (op.kind == OP_HEAD && op_is_noop(e, op, n)) {
        for (size_t k = 0; k < e->ops.size(); ++k) if (e->ops[k].kind == OP_HEAD && e->ops[k].level == 2) return (int)k;
    }
    return i;
}

// channel ranges an op reads / writes, as (buffer, first channel, channels, pixels per frame, scale) records
struct IoView { int buf, co, C; double px, scale; };
static void op_reads(const zly_engine* e, const Op& op, std::vector<IoView>* out)
{
    if (op.kind == OP_CONV) {
        const Buffer& ib = e->bufs[(size_t)op.in.buf];
        out->push_back({op.in.buf, op.in.co, op.in.C, (double)ib.H * ib.W, op.cin_frac});
        if (op.in2.buf >= 0) { const Buffer& b2 = e->bufs[(size_t)op.in2.buf]; out->push_back({op.in2.buf, op.in2.co, op.in2.C, (double)b2.H * b2.W, op.cin_frac}); }
        if (op.res.buf >= 0) { const Buffer& rb = e->bufs[(size_t)op.res.buf]; out->push_back({op.res.buf, op.res.co, op.res.C > 0 ? op.res.C : op.cout, (double)rb.H * rb.W, 1.0}); }
    } else if (op.kind == OP_SPPF) {
        const Buffer& b = e->bufs[(size_t)op.in.buf];
        out->push_back({op.in.buf, op.in.co, op.c, (double)b.H * b.W, 1.0});
    } else if (op.kind == OP_HEAD) {                      // a tail op stands for its own level; one launch for all levels = a group of the three
        const int l = op.level;
        const Buffer& bb = e->bufs[(size_t)op.head_box[l]];
        const Buffer& cb = e->bufs[(size_t)op.head_cls[l]];
        out->push_back({op.head_box[l], 0, bb.C, (double)bb.H * bb.W, 1.0});
        out->push_back({op.head_cls[l], 0, cb.C, (double)cb.H * cb.W, 1.0});
    }
}
static void op_writes(const zly_engine* e, const Op& op, std::vector<IoView>* out)
{
    if (op.kind == OP_CONV) {
        const Buffer& ob = e->bufs[(size_t)op.out.buf];
        out->push_back({op.out.buf, op.out.co, op.cout, (double)ob.H * ob.W, 1.0});
    } else if (op.kind == OP_SPPF) {
        const Buffer& b = e->bufs[(size_t)op.in.buf];
        out->push_back({op.in.buf, op.in.co + op.c, 3 * op.c, (double)b.H * b.W, 1.0});
    } else if (op.kind == OP_PREPROCESS) {
        const Buffer& b = e->bufs[(size_t)e->in_buf];
        out->push_back({e->in_buf, 0, 8, (double)b.H * b.W, 3.0 / 8.0});
    }
}

static hipError_t run_op(zly_engine* e, const Op& op, int n, const uint8_t* d_src, void* d_slabs_out, uint32_t tag0, hipStream_t s)
{
    switch (op.kind) {
    case OP_PREPROCESS:
        return launch_preprocess(e->dtype, d_src, e->d_desc, n, e->bufs[(size_t)e->in_buf].ptr, nullptr, e->cfg.model_w, e->cfg.model_h, s);
    case OP_CONV: {
        const Buffer& ib = e->bufs[(size_t)op.in.buf];
        if (c2f_covered(e, op, n)) return hipSuccess;               // computed by the fused C2f kernel launched at its leader
        if (sppf_active(e)) {
            const int oi = (int)(&op - e->ops.data());
            if (oi == e->sppf_cv2) return hipSuccess;               // computed by the fused SPPF kernel launched at model.9.cv1
            if (oi == e->sppf_cv1) {
                const Op& o2 = e->ops[(size_t)e->sppf_cv2];
                const Buffer& cb = e->bufs[(size_t)op.out.buf];     // the block's concat buffer [y | p1 | p2 | p3]
                const Buffer& ob2 = e->bufs[(size_t)o2.out.buf];
                const char* wb = (const char*)e->d_weights;
                SppfArgs sa{};
                sa.x = ib.ptr; sa.x_cs = ib.C; sa.x_co = op.in.co; sa.Cin = op.in.C;
                sa.w1 = wb + op.w_off; sa.b1 = (const float*)(wb + op.b_off);
                sa.w2 = wb + o2.w_off; sa.b2 = (const float*)(wb + o2.b_off);
                sa.out = ob2.ptr; sa.out_cs = ob2.C; sa.out_co = o2.out.co; sa.Cout = o2.cout;
                sa.cat = cb.ptr; sa.cat_cs = cb.C;
                sa.H = cb.H; sa.W = cb.W; sa.n = n; sa.c = op.cout; sa.split = 0;
                sa.dump = (e->cfg.flags & ZLY_FLAG_DUMP_LOGITS) ? 1 : 0;
                return launch_sppf_fused(sa, s);
            }
        }
        if (detect_merge_active(e, n)) {
            const int role = detect_merge_role(e, (int)(&op - e->ops.data()));
            if (role == 1) return hipSuccess;
            if (role == 2 || role == 3) {
                ConvArgsMulti m{};
                for (int l = 0; l < 3; ++l) {
                    if (role == 2) m.a[m.n++] = make_conv_args(e, e->ops[(size_t)e->det_stem[l]], n);
                    else { m.a[m.n++] = make_conv_args(e, e->ops[(size_t)e->det_a[l]], n); m.a[m.n++] = make_conv_args(e, e->ops[(size_t)e->det_b[l]], n); }
                }
                return launch_conv_multi(m, role == 2 ? 3 : 2, s);
            }
        }
        if (const C2fPlan* pl = c2f_active(e, op, n)) {
            const Buffer& cb = e->bufs[(size_t)op.c2f_cat];
            const Buffer& xb = e->bufs[(size_t)op.c2f_x.buf];
            const Buffer& ob2 = e->bufs[(size_t)op.c2f_out.buf];
            const char* wb = (const char*)e->d_weights;
            C2fArgs ca{};
            ca.x = xb.ptr; ca.x_cs = xb.C; ca.x_co = op.c2f_x.co;
            if (op.c2f_x2.buf >= 0) {
                const Buffer& x2 = e->bufs[(size_t)op.c2f_x2.buf];
                ca.x2 = x2.ptr; ca.x2_cs = x2.C; ca.x2_co = op.c2f_x2.co; ca.split_c = op.c2f_x.C;
            }
            ca.w1 = wb + op.c2f_w1; ca.b1 = (const float*)(wb + op.c2f_b1); ca.nk1 = op.c2f_nk1;
            ca.cat = cb.ptr; ca.cat_cs = cb.C; ca.pair_in_co = op.c2f_in_co; ca.pair_out_co = op.c2f_out_co;
            ca.wA = wb + op.c2f_wA; ca.bA = (const float*)(wb + op.c2f_bA); ca.wB = wb + op.c2f_wB; ca.bB = (const float*)(wb + op.c2f_bB);
            ca.res = op.c2f_res;
            ca.w2 = wb + op.c2f_w2; ca.b2 = (const float*)(wb + op.c2f_b2); ca.nk2 = op.c2f_nk2; ca.Cout2 = op.c2f_cout2;
            ca.out = ob2.ptr; ca.out_cs = ob2.C; ca.out_co = op.c2f_out.co;
            ca.H = cb.H; ca.W = cb.W; ca.n = n;
            ca.TH = pl->th; ca.TW = pl->tw; ca.tiles_x = pl->tiles_x; ca.tiles_y = pl->tiles_y; ca.total_tiles = pl->total_tiles;
            ca.dump = (e->cfg.flags & ZLY_FLAG_DUMP_LOGITS) ? 1 : 0;
            if (op.c2f_c == 64 && op.c2f_mid.buf >= 0) { const Buffer& mb = e->bufs[(size_t)op.c2f_mid.buf]; ca.mid = mb.ptr; ca.mid_cs = mb.C; }
            return launch_c2f(op.c2f_c, op.c2f_mode, ca, *pl, s);
        }
        if (op.pair) {
            const PairPlan* pl = pair_active(e, op, n);
            if (pl && op.pair == 2) return hipSuccess;             // computed by the pair kernel launched at the first conv
            if (pl) {
                const Buffer& pb = e->bufs[(size_t)op.pair_out.buf];
                PairArgs pa;
                pa.in = ib.ptr; pa.in_cs = ib.C; pa.in_co = op.in.co;
                pa.out = pb.ptr; pa.out_cs = pb.C; pa.out_co = op.pair_out.co;
                pa.wA = (const char*)e->d_weights + op.pair_wA; pa.bA = (const float*)((const char*)e->d_weights + op.pair_bA);
                pa.wB = (const char*)e->d_weights + op.pair_wB; pa.bB = (const float*)((const char*)e->d_weights + op.pair_bB);
                pa.H = ib.H; pa.W = ib.W; pa.n = n;
                pa.TH = pl->th; pa.TW = pl->tw; pa.tiles_x = pl->tiles_x; pa.tiles_y = pl->tiles_y; pa.total_tiles = pl->total_tiles;
                pa.res = op.pair_res;
                return launch_pair(op.pair_c, pa, *pl, s);
            }
        }
        ConvArgs a = make_conv_args(e, op, n);
        if (wsk_active(e, op, n)) {
            ConvArgs k = a;
            k.Cin = 80; k.K = 9 * 80; k.nk = 23; k.cout_pad = 80;
            k.wgt = (const char*)e->d_weights + op.wsk_w;
            const hipError_t r = launch_conv_wsk(k, s);
            if (r != hipErrorInvalidValue) return r;             // a tensor beyond the kernel's 32-bit offsets takes the generic path below
        }
        return launch_conv(e->dtype, a, conv_launch_of(e, op, n), s);
    }
    case OP_SPPF: {
        if (sppf_active(e)) return hipSuccess;                       // inside the fused SPPF kernel
        const Buffer& b = e->bufs[(size_t)op.in.buf];
        return launch_sppf_pool(e->dtype, b.ptr, b.C, op.c, n, b.H, b.W, s, e->sw.pool_six_pass ? 1 : 0);
    }
    case OP_HEAD: {
        HeadArgs h = op.head;
        if (lanes_active(e, n) && !e->sw.no_tail_split) h.only_level = op.level;       // per-level launches (+0.3 % at batch 64)
        else if (op.level != 2) return hipSuccess;          // the last tail op covers all levels
        else h.only_level = -1;
        h.head = (e->cfg.flags & ZLY_FLAG_NO_HEAD_TENSOR) ? nullptr : e->d_head; h.desc = e->d_desc; h.conf_thr = e->cfg.conf_thr; h.cand = e->cur_cand; h.cand_count = e->cur_count;
        return launch_head_fused(e->dtype, h, n, s);
    }
    case OP_NMS:
        return launch_nms(e->cur_cand, e->cur_count, e->N, n, e->cfg.iou_thr, e->nc, e->d_scratch,
                          d_slabs_out ? d_slabs_out : e->d_slabs, e->cfg.max_dets, tag0, s, e->sw.nms_general ? 1 : 0);
    }
    return hipErrorInvalidValue;
}

// ops [first, last) -- the graph-capturable middle of the path is [1, ops.size()-1).  Side-lane ops are
// launched on the engine's side streams behind a fork event; the Detect tail (OP_HEAD) joins them.
static hipError_t run_ops(zly_engine* e, size_t first, size_t last, int n, const uint8_t* d_src, void* d_slabs_out, uint32_t tag0, hipStream_t s)
{
    bool forked[3] = {false, false, false};
    hipError_t r = hipSuccess;
    auto join_all = [&]() {
        for (int l = 1; l <= 2 && r == hipSuccess; ++l) {
            if (!forked[l]) continue;
            r = hipEventRecord(e->ev_join[l - 1], e->side[l - 1]);
            if (r == hipSuccess) r = hipStreamWaitEvent(s, e->ev_join[l - 1], 0);
            forked[l] = false;
        }
    };
    for (size_t i = first; i < last && r == hipSuccess; ++i) {
        const Op& op = e->ops[i];
        hipStream_t st = s;
        // measured: at batch 1 the cross-stream edges cost more than the overlap buys (0.27 -> 0.34 ms/frame);
        // from batch 16 up the Detect branches are long enough to pay (1.42 -> 1.38 ms per 64 frames)
        if (op.lane > 0 && lanes_active(e, n)) {
            st = e->side[op.lane - 1];
            if (!forked[op.lane]) {
                r = hipEventRecord(e->ev_fork[op.lane - 1], s);
                if (r == hipSuccess) r = hipStreamWaitEvent(st, e->ev_fork[op.lane - 1], 0);
                forked[op.lane] = true;
                if (r != hipSuccess) break;
            }
        }
        if (op.kind == OP_NMS) { join_all(); if (r != hipSuccess) break; }       // the tail launches only append candidates: no join needed before them
        // the side streams are joined in front of the P5 branch rather than at the very end: they have long finished by then, and
        // the hand-over (~10 us on the main stream) then overlaps nothing less than at the step boundary (+0.7 %)
        if (op.kind == OP_CONV && op.lane == 0 && op.name.rfind("model.22.cv2.2.0", 0) == 0) { join_all(); if (r != hipSuccess) break; }
        // ZLY_ABLATE_SKIP=<op name>[,<op name>...]: the named launches are left out (results are garbage): measures what a launch costs the
        // step when several engines' chains overlap, which its isolated duration does not tell (tools/ablate_launches.sh)
#ifdef ZLY_DIAG         // diagnostic build only (libzly_diag.so): a shipped engine cannot be told to skip launches
        if (!e->sw.ablate.empty() && e->sw.ablate.find("," + op.name + ",") != std::string::npos) continue;
#endif
        r = run_op(e, op, n, d_src, d_slabs_out, tag0, st);
    }
    if (r == hipSuccess) join_all();
    return r;
}

// every stream that may hold an outstanding deferred NMS is joined into `s`
// lag = 0: all of them; lag = 1: all but the most recent call's (so that the caller can consume call k-1's slabs on
// `s` right after enqueuing call k without putting NMS(k) in front of call k+1).  NMS launches are serialised on one
// stream, so waiting for a call's event covers every earlier call.
static int join_nms(zly_engine* e, hipStream_t s, int lag = 0)
{
    if (e->last_async < 0) return ZLY_OK;                // nothing outstanding
    const int p = lag == 0 ? e->last_async : e->last_async ^ 1;
    if (e->nms_recorded[p]) HIP_TRY(hipStreamWaitEvent(s, e->ev_nms[p], 0), ZLY_ERR_INFERENCE);
    return ZLY_OK;
}

// The process-wide enqueue gate.  Enqueue sections of different engines (launches, async copies, event records: ~0.03-0.1 ms of host
// work per batch) hold it SHARED and run concurrently -- with ZLY_NUM_DEVICES x ZLY_ENGINES_PER_GPU dispatcher threads a single mutex
// here serialised sixteen threads at the same order of time a GPU needs per batch.  What must be alone in the process holds it
// EXCLUSIVE: stream capture (a HIP call from another engine's thread while a capture was open -- a launch, an async copy, an event
// wait -- failed both sides with "operation failed due to a previous error during capture", thread-local and relaxed modes alike) and
// device allocation / release (zly_create, zly_destroy, the staging ring: hot reload builds engines beside running ones).  Captures
// happen at zly_create (batch 1 and max_batch) and otherwise only on the first call of a new batch size on the synchronous entry points.
// Writers are preferred: a waiting capture is not starved by a steady stream of enqueues.
class EnqueueGate {
    std::mutex m_;
    std::condition_variable cv_;
    int readers_ = 0, writers_waiting_ = 0;
    bool writer_ = false;
public:
    void lock_shared() { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return !writer_ && writers_waiting_ == 0; }); ++readers_; }
    void unlock_shared() { std::lock_guard<std::mutex> lk(m_); if (--readers_ == 0) cv_.notify_all(); }
    void lock() { std::unique_lock<std::mutex> lk(m_); ++writers_waiting_; cv_.wait(lk, [&] { return !writer_ && readers_ == 0; }); --writers_waiting_; writer_ = true; }
    void unlock() { std::lock_guard<std::mutex> lk(m_); writer_ = false; cv_.notify_all(); }
};
static EnqueueGate g_gate;
struct SharedGate {
    bool held = true;
    SharedGate() { g_gate.lock_shared(); }
    ~SharedGate() { if (held) g_gate.unlock_shared(); }
    void release() { if (held) { g_gate.unlock_shared(); held = false; } }
    SharedGate(const SharedGate&) = delete; SharedGate& operator=(const SharedGate&) = delete;
};
struct ExclusiveGate {
    ExclusiveGate() { g_gate.lock(); }
    ~ExclusiveGate() { g_gate.unlock(); }
    ExclusiveGate(const ExclusiveGate&) = delete; ExclusiveGate& operator=(const ExclusiveGate&) = delete;
};

// phase timing of the last sampled call, if its events have completed: added to the stats, never waited for
static void harvest_timing(zly_engine* e)
{
    if (!e->t_pending || hipEventQuery(e->ev_t[3]) != hipSuccess) return;
    float pre = 0.f, fwd = 0.f, post = 0.f;
    if (hipEventElapsedTime(&pre, e->ev_t[0], e->ev_t[1]) == hipSuccess && hipEventElapsedTime(&fwd, e->ev_t[1], e->ev_t[2]) == hipSuccess &&
        hipEventElapsedTime(&post, e->ev_t[2], e->ev_t[3]) == hipSuccess) {
        const int nf = e->t_frames;
        with_stats(e, [&](zly_stats& st) {
            st.sampled_frames += (uint64_t)nf;
            st.sampled_preprocess_ms += pre; st.sampled_forward_ms += fwd; st.sampled_postprocess_ms += post > 0.f ? post : 0.f;
        });
    }
    e->t_pending = false;
}

// nms_stream_out: the stream the call's NMS (its last kernel) was launched on -- `s`, or the engine's NMS stream when deferred
static int run_path(zly_engine* e, int n, const uint8_t* d_src, void* d_slabs_out, uint32_t tag0, hipStream_t s, bool with_pre, bool defer_nms = false,
                    hipStream_t* nms_stream_out = nullptr)
{
    const size_t nops = e->ops.size();
    harvest_timing(e);
    const bool sample = with_pre && !e->t_pending && (e->sample_ctr++ % zly_engine::SAMPLE_EVERY) == 0;
    with_stats(e, [&](zly_stats& st) { st.batches++; });
    // like the Detect side streams: below batch 16 the cross-stream edges (event record + two stream waits per call) cost
    // more than the overlap buys -- measured 0.236 -> 0.290 ms per frame at batch 1 -- so small batches stay in stream order
    defer_nms = defer_nms && e->nms_stream != nullptr && n >= 16;
    int par = 0;
    if (defer_nms) {
        // this call's Detect tail fills candidate buffer `par`; the NMS that last read it (two calls back) must be done
        par = e->parity;
        if (e->nms_recorded[par]) HIP_TRY(hipStreamWaitEvent(s, e->ev_nms[par], 0), ZLY_ERR_INFERENCE);
    } else if (e->last_async >= 0) {
        int rcj = join_nms(e, s);                       // a synchronous-path call after deferred ones: plain stream order again
        if (rcj != ZLY_OK) return rcj;
        e->last_async = -1;
    }
    e->cur_cand = par ? e->d_cand_alt : e->d_cand;
    e->cur_count = par ? e->d_count_alt : e->d_count;
    // ops[0] = preprocess, ops[1] = model.0.  On the detect paths of the bf16 engine both are ONE kernel
    // (kernels_stem.hip); zly_forward (caller-supplied fp32 images) keeps the generic model.0 conv.
    const bool fused = with_pre && e->stem_fused;
    size_t first = 1;
    if (sample) HIP_TRY(hipEventRecord(e->ev_t[0], s), ZLY_ERR_INFERENCE);
    e->last_stem1 = fused && e->stem1;
    if (fused && e->stem1) {
        Stem1Args st = e->stem1a;
        st.st.src = d_src; st.st.desc = e->d_desc;
#ifdef ZLY_DIAG
        if (e->sw.ablate.find(",stem,") != std::string::npos) { /* ZLY_ABLATE_SKIP=stem (libzly_diag.so only) */ } else
#endif
        HIP_TRY(launch_stem_model1(st, n, s), ZLY_ERR_INFERENCE);
        first = 3;
    } else if (fused) {
        StemArgs st = e->stem;
        st.src = d_src; st.desc = e->d_desc;
        HIP_TRY(launch_stem_fused(st, n, s), ZLY_ERR_INFERENCE);
        first = 2;
    } else if (with_pre) {
        HIP_TRY(run_op(e, e->ops[0], n, d_src, nullptr, 0, s), ZLY_ERR_INFERENCE);
    }
    if (sample) HIP_TRY(hipEventRecord(e->ev_t[1], s), ZLY_ERR_INFERENCE);
    // Graph replay.  Graphs are captured at zly_create for batch 1 and max_batch (warm_batch) and, on the synchronous entry points, on the
    // first call of any other batch size.  The pipelined host path (dispatcher threads, several per process) never captures: it replays
    // what exists -- a lone frame and a full batch, its two steady states -- and launches other partial batches eagerly with cached
    // kernel shapes: no capture storm over the 62 partial sizes, no capture while other engines run.
    const int key = (n * 2 + (fused ? 1 : 0)) * 2 + par;     // the captured Detect tail holds the candidate buffer's address
    auto git = e->graphs.find(key);
    if (e->cfg.use_graph && git == e->graphs.end() && !e->ingest_active) {
        // Capture on the engine's own stream, then replay on whichever stream the caller uses.  The caller holds the enqueue gate shared;
        // a capture needs it exclusively (see EnqueueGate).  e->mu is held throughout, so nothing else touches this engine meanwhile.
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        HIP_TRY(hipStreamSynchronize(s), ZLY_ERR_INFERENCE);
        g_gate.unlock_shared();
        {
            ExclusiveGate x;
            hipError_t r = hipStreamBeginCapture(e->stream, hipStreamCaptureModeRelaxed);
            if (r == hipSuccess) {
                r = run_ops(e, first, nops - 1, n, nullptr, nullptr, 0, e->stream);
                hipError_t r2 = hipStreamEndCapture(e->stream, &g);
                if (r == hipSuccess) r = r2;
            }
            if (r == hipSuccess) r = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            if (g) hipGraphDestroy(g);
            if (r != hipSuccess) { (void)hipGetLastError(); ge = nullptr; }
        }
        g_gate.lock_shared();
        // a failed capture is retried once (on the next call of this shape) before the shape is left on eager launches for good
        if (ge || ++e->graph_failures[key] >= 2) git = e->graphs.emplace(key, ge).first;
    }
    if (e->cfg.use_graph && git != e->graphs.end() && git->second) {
        HIP_TRY(hipGraphLaunch(git->second, s), ZLY_ERR_INFERENCE);
        with_stats(e, [](zly_stats& st) { st.graph_replays++; });
    } else {
        HIP_TRY(run_ops(e, first, nops - 1, n, nullptr, nullptr, 0, s), ZLY_ERR_INFERENCE);
        with_stats(e, [](zly_stats& st) { st.eager_batches++; });
    }
    if (sample) HIP_TRY(hipEventRecord(e->ev_t[2], s), ZLY_ERR_INFERENCE);
    hipStream_t ns = s;
    if (defer_nms) {
        ns = e->nms_stream;
        HIP_TRY(hipEventRecord(e->ev_head, s), ZLY_ERR_INFERENCE);
        HIP_TRY(hipStreamWaitEvent(e->nms_stream, e->ev_head, 0), ZLY_ERR_INFERENCE);
        HIP_TRY(run_op(e, e->ops[nops - 1], n, nullptr, d_slabs_out, tag0, e->nms_stream), ZLY_ERR_INFERENCE);
        HIP_TRY(hipEventRecord(e->ev_nms[par], e->nms_stream), ZLY_ERR_INFERENCE);
        e->nms_recorded[par] = true; e->last_async = par; e->parity = par ^ 1;
    } else {
        HIP_TRY(run_op(e, e->ops[nops - 1], n, nullptr, d_slabs_out, tag0, s), ZLY_ERR_INFERENCE);
    }
    if (sample) {
        HIP_TRY(hipEventRecord(e->ev_t[3], ns), ZLY_ERR_INFERENCE);
        e->t_pending = true; e->t_frames = n;
    }
    if (nms_stream_out) *nms_stream_out = ns;
    e->last_n = n;
    return ZLY_OK;
}

static int set_desc(zly_engine* e, int n, const int32_t* w, const int32_t* h, const size_t* offs, hipStream_t s)
{
    bool same = (int)e->desc_cache.size() >= n;
    for (int i = 0; i < n && same; ++i)
        same = e->desc_cache[(size_t)i].w == w[i] && e->desc_cache[(size_t)i].h == h[i] && e->desc_cache[(size_t)i].src_off == offs[i];
    if (same) return ZLY_OK;
    // frame sizes / offsets changed: next entry of the pinned ring, uploaded in stream order.  Calls on one engine are
    // stream-ordered by contract (they share every activation buffer), so the upload cannot overtake a kernel of the
    // previous call that still reads d_desc; the ring entry itself is only reused once ITS upload has completed.
    const int r = e->desc_next;
    e->desc_next = (r + 1) % zly_engine::DESC_RING;
    if (e->desc_used[r]) HIP_TRY(hipEventSynchronize(e->ev_desc[r]), ZLY_ERR_INFERENCE);     // 8 uploads back: long complete
    FrameDesc* hd = e->h_desc + (size_t)r * (size_t)e->cfg.max_batch;
    if ((int)e->desc_cache.size() < n) e->desc_cache.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        FrameDesc d; d.src_off = offs[i]; d.w = w[i]; d.h = h[i];
        hd[i] = d;
        e->desc_cache[(size_t)i] = d;
    }
    for (size_t i = (size_t)n; i < e->desc_cache.size(); ++i) e->desc_cache[i].w = -1;
    HIP_TRY(hipMemcpyAsync(e->d_desc, hd, sizeof(FrameDesc) * (size_t)n, hipMemcpyHostToDevice, s), ZLY_ERR_INFERENCE);
    HIP_TRY(hipEventRecord(e->ev_desc[r], s), ZLY_ERR_INFERENCE);
    e->desc_used[r] = true;
    return ZLY_OK;
}

static int ensure_stage(zly_engine* e, size_t bytes)
{
    if (bytes <= e->stage_bytes) return ZLY_OK;
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_SYSTEM);
    if (e->d_stage) hipFree(e->d_stage);
    if (e->h_stage) hipHostFree(e->h_stage);
    e->d_stage = nullptr; e->h_stage = nullptr; e->stage_bytes = 0;
    const size_t cap = (bytes + (1u << 20) - 1) / (1u << 20) * (1u << 20);
    HIP_TRY(hipMalloc((void**)&e->d_stage, cap), ZLY_ERR_SYSTEM);
    HIP_TRY(hipHostMalloc((void**)&e->h_stage, cap, hipHostMallocDefault), ZLY_ERR_SYSTEM);
    e->stage_bytes = cap;
    return ZLY_OK;
}

static int ensure_scratch_f32(zly_engine* e, size_t elems)
{
    if (elems <= e->scratch_f32_elems) return ZLY_OK;
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_SYSTEM);
    if (e->d_scratch_f32) hipFree(e->d_scratch_f32);
    e->d_scratch_f32 = nullptr; e->scratch_f32_elems = 0;
    HIP_TRY(hipMalloc((void**)&e->d_scratch_f32, elems * sizeof(float)), ZLY_ERR_SYSTEM);
    e->scratch_f32_elems = elems;
    return ZLY_OK;
}

static uint64_t now_ms()
{
    return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

static void ingest_destroy(zly_engine* e);

// Returns the first HIP error met while draining / releasing (the engine is gone either way).
static hipError_t destroy_engine(zly_engine* e)
{
    if (!e) return hipSuccess;
    hipError_t first = hipSuccess;
    auto note = [&](hipError_t r) { if (r != hipSuccess && first == hipSuccess) first = r; };
    note(hipSetDevice(e->dev));
    ingest_destroy(e);                                   // joins the dispatcher / completion threads (they take the gate themselves)
    // drain this engine's streams with no gate held (up to a batch of device time: other engines keep enqueuing) ...
    if (e->stream) note(hipStreamSynchronize(e->stream));
    for (int i = 0; i < 2; ++i) if (e->side[i]) note(hipStreamSynchronize(e->side[i]));
    if (e->nms_stream) note(hipStreamSynchronize(e->nms_stream));
    // ... and release alone in the process: a hipFree / hipGraphExecDestroy from this thread while another engine (a hot reload builds the
    // new engines beside the running ones) has a capture open fails both sides
    ExclusiveGate x;
    for (auto& kv : e->graphs) if (kv.second) note(hipGraphExecDestroy(kv.second));
    for (Buffer& b : e->bufs) if (b.ptr) note(hipFree(b.ptr));
    void* dptrs[] = {e->d_weights, e->d_head, e->d_cand, e->d_cand_alt, e->d_count_alt, e->d_scratch, e->d_count, e->d_slabs, e->d_desc, e->d_stage, e->d_scratch_f32};
    for (void* p : dptrs) if (p) note(hipFree(p));
    if (e->h_desc) note(hipHostFree(e->h_desc));
    if (e->h_stage) note(hipHostFree(e->h_stage));
    if (e->h_slabs) note(hipHostFree(e->h_slabs));
    for (int i = 0; i < 2; ++i) {
        if (e->ev_fork[i]) hipEventDestroy(e->ev_fork[i]);
        if (e->ev_join[i]) hipEventDestroy(e->ev_join[i]);
        if (e->side[i]) hipStreamDestroy(e->side[i]);
    }
    if (e->ev_head) hipEventDestroy(e->ev_head);
    for (int i = 0; i < 2; ++i) if (e->ev_nms[i]) hipEventDestroy(e->ev_nms[i]);
    for (int i = 0; i < zly_engine::DESC_RING; ++i) if (e->ev_desc[i]) hipEventDestroy(e->ev_desc[i]);
    for (int i = 0; i < 4; ++i) if (e->ev_t[i]) hipEventDestroy(e->ev_t[i]);
    for (int i = 0; i < 2; ++i) if (e->ev_call[i]) hipEventDestroy(e->ev_call[i]);
    if (e->nms_stream) hipStreamDestroy(e->nms_stream);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
    return first;
}


// ------------------------------------------------------------------------------------------------
// pipelined host-to-host path: zly_submit / zly_poll / zly_wait  (include/zly.h)
//
//   submitting threads --memcpy--> pinned slot k+2      (host cores, in parallel)
//   copy stream        --H2D-----> d_stage of slot k+1  (PCIe, while ...)
//   engine stream      -- path --> slot k               (... the previous batch computes)
//   d2h stream         --D2H-----> pinned slabs of slot k-1, completion thread wakes the waiters
//
// A ring slot goes FREE -> OPEN (accepting frames) -> CLOSED (no more frames; copies may still be in progress) ->
// INFLIGHT (enqueued on the device) -> DONE (results in pinned host memory) -> FREE (every ticket consumed).  Slots are
// used in ring order, so batch b lives in slot b % S and a ticket is (batch << 16 | index in batch).
// ------------------------------------------------------------------------------------------------
enum { SLOT_FREE = 0, SLOT_OPEN, SLOT_CLOSED, SLOT_INFLIGHT, SLOT_DONE };

struct IngestSlot {
    uint8_t* h_stage = nullptr; uint8_t* d_stage = nullptr;
    unsigned char* h_slabs = nullptr; unsigned char* d_slabs = nullptr;
    hipEvent_t ev_h2d = nullptr, ev_done = nullptr, ev_out = nullptr;
    int state = SLOT_FREE;
    uint64_t batch = 0;
    int n_reserved = 0, n_committed = 0, n_consumed = 0;
    size_t bytes_used = 0;
    std::vector<int32_t> w, h;
    std::vector<size_t> off;
    std::vector<uint8_t> consumed;
    int rc = ZLY_OK;
    std::string err;
    uint64_t ts_ms = 0;
};

struct Ingest {
    std::mutex mu;
    std::condition_variable cv_free;     // submitters waiting for a slot to open
    std::condition_variable cv_disp;     // dispatcher: frames arrived / copies finished / a batch completed
    std::condition_variable cv_comp;     // completion thread: a batch went in flight
    std::condition_variable cv_done;     // waiters: a batch completed
    std::vector<IngestSlot> slots;
    size_t slot_bytes = 0;
    int open = -1;                       // slot accepting frames, -1 = none
    uint64_t next_batch = 0;             // batch number the next opened slot gets (slot = batch % S)
    std::deque<int> closed, inflight;
    int depth = 2;                       // batches enqueued on the device at once
    bool stop = false;
    std::thread dispatcher, completer;
    hipStream_t copy_stream = nullptr, d2h_stream = nullptr;
    bool use_d2h_stream = false;
};

// One H2D and one D2H stream per device, shared by every engine of the process: each HIP stream beyond the hardware-queue limit (4 by default)
// shares a hardware queue with another stream and the two serialise -- an upload stream of engine B landing on the queue of engine A's
// compute stream stalled A for a whole 33 MB transfer.  Uploads of different engines share the PCIe link anyway.
static hipStream_t g_h2d_stream[16] = {}, g_d2h_stream[16] = {};

static int env_int(const char* name, int fallback) { const char* v = getenv(name); return (v && *v) ? atoi(v) : fallback; }

// under ing->mu: make the next ring slot the open one if it is free
static bool ingest_try_open(zly_engine* e, Ingest* g)
{
    IngestSlot& sl = g->slots[(size_t)(g->next_batch % g->slots.size())];
    if (sl.state != SLOT_FREE) return false;
    sl.state = SLOT_OPEN; sl.batch = g->next_batch++;
    sl.n_reserved = sl.n_committed = sl.n_consumed = 0; sl.bytes_used = 0; sl.rc = ZLY_OK; sl.err.clear();
    std::fill(sl.consumed.begin(), sl.consumed.end(), (uint8_t)0);
    g->open = (int)(sl.batch % g->slots.size());
    (void)e;
    return true;
}

static void ingest_close_open(Ingest* g)
{
    if (g->open < 0) return;
    g->slots[(size_t)g->open].state = SLOT_CLOSED;
    g->closed.push_back(g->open);
    g->open = -1;
}

// enqueue one closed, fully copied batch: H2D on the copy stream, the path on the engine's stream, slabs back on the d2h stream
static void ingest_enqueue(zly_engine* e, Ingest* g, IngestSlot& sl)
{
    std::lock_guard<std::mutex> lk(e->mu);
    SharedGate gl;
    const int n = sl.n_reserved;
    hipStream_t out_stream = e->stream;
    auto body = [&]() -> int {
        HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
        HIP_TRY(hipMemcpyAsync(sl.d_stage, sl.h_stage, sl.bytes_used, hipMemcpyHostToDevice, g->copy_stream), ZLY_ERR_INFERENCE);
        HIP_TRY(hipEventRecord(sl.ev_h2d, g->copy_stream), ZLY_ERR_INFERENCE);
        HIP_TRY(hipStreamWaitEvent(e->stream, sl.ev_h2d, 0), ZLY_ERR_INFERENCE);
        int rc = set_desc(e, n, sl.w.data(), sl.h.data(), sl.off.data(), e->stream);
        if (rc != ZLY_OK) return rc;
        hipStream_t ns = e->stream;
        e->ingest_active = true;
        rc = run_path(e, n, sl.d_stage, sl.d_slabs, (uint32_t)(sl.batch << 16), e->stream, true, (e->cfg.flags & ZLY_FLAG_ASYNC_NMS) != 0, &ns);
        e->ingest_active = false;
        if (rc != ZLY_OK) return rc;
        // the slabs (n x 2.6 KB) go back on the stream the NMS ran on, right behind it: a download stream of its own would be the fifth
        // stream of a three-engine process and share a hardware queue with a compute stream (ZLY_D2H_STREAM=1 restores it)
        out_stream = g->use_d2h_stream ? g->d2h_stream : ns;
        if (g->use_d2h_stream) {
            HIP_TRY(hipEventRecord(sl.ev_done, ns), ZLY_ERR_INFERENCE);
            HIP_TRY(hipStreamWaitEvent(g->d2h_stream, sl.ev_done, 0), ZLY_ERR_INFERENCE);
        }
        HIP_TRY(hipMemcpyAsync(sl.h_slabs, sl.d_slabs, slab_bytes_of(e) * (size_t)n, hipMemcpyDeviceToHost, out_stream), ZLY_ERR_INFERENCE);
        return ZLY_OK;
    };
    sl.rc = body();
    if (sl.rc != ZLY_OK) sl.err = g_last_error;
    hipEventRecord(sl.ev_out, out_stream);             // also on failure: the completion thread must never wait forever
}

static void ingest_dispatch_loop(zly_engine* e, Ingest* g)
{
    hipSetDevice(e->dev);
    std::unique_lock<std::mutex> lk(g->mu);
    while (true) {
        // work = a closed batch, or frames in the open batch while the device has room (no batching window: a lone frame goes at once)
        g->cv_disp.wait(lk, [&] {
            if (g->stop) return true;
            if ((int)g->inflight.size() >= g->depth) return false;
            if (!g->closed.empty()) return g->slots[(size_t)g->closed.front()].n_committed == g->slots[(size_t)g->closed.front()].n_reserved;
            return g->open >= 0 && g->slots[(size_t)g->open].n_reserved > 0;
        });
        if (g->stop) return;
        if (g->closed.empty()) {
            ingest_close_open(g);
            ingest_try_open(e, g);
            g->cv_free.notify_all();
            continue;                                  // re-evaluate: its copies may still be running
        }
        const int si = g->closed.front();
        g->closed.pop_front();
        IngestSlot& sl = g->slots[(size_t)si];
        lk.unlock();
        ingest_enqueue(e, g, sl);
        lk.lock();
        sl.state = SLOT_INFLIGHT;
        g->inflight.push_back(si);
        g->cv_comp.notify_one();
    }
}

static void ingest_complete_loop(zly_engine* e, Ingest* g)
{
    hipSetDevice(e->dev);
    std::unique_lock<std::mutex> lk(g->mu);
    while (true) {
        g->cv_comp.wait(lk, [&] { return g->stop || !g->inflight.empty(); });
        if (g->inflight.empty()) { if (g->stop) return; continue; }
        const int si = g->inflight.front();
        IngestSlot& sl = g->slots[(size_t)si];
        lk.unlock();
        const hipError_t r = hipEventSynchronize(sl.ev_out);
        const uint64_t ts = now_ms();                                       // onnx_engine.cpp:813-815
        lk.lock();
        if (r != hipSuccess && sl.rc == ZLY_OK) { sl.rc = ZLY_ERR_INFERENCE; sl.err = std::string("hipEventSynchronize: ") + hipGetErrorString(r); }
        sl.ts_ms = ts;
        sl.state = SLOT_DONE;
        g->inflight.pop_front();
        const int n = sl.n_reserved;
        const bool ok = sl.rc == ZLY_OK;
        with_stats(e, [&](zly_stats& st) { if (ok) st.inference_count += (uint64_t)n; else st.inference_errors += (uint64_t)n; });
        g->cv_done.notify_all();
        g->cv_disp.notify_one();
    }
}

static void ingest_free(zly_engine* e, Ingest* g);

// first zly_submit: allocate the ring and start the two engine-owned threads (under e->mu)
static int ingest_start(zly_engine* e)
{
    std::lock_guard<std::mutex> lk(e->mu);
    ExclusiveGate gl;                                  // allocations: alone in the process (another engine may otherwise have a capture open)
    if (e->ingest.load()) return ZLY_OK;
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    Ingest* g = new Ingest();
    // six slots (round 4; four before): two batches on the device, one filling, and three whose results wait for their consumer.  A host that delivers results
    // in submission order across several engines (the plugin) consumes an engine's finished batch only when the other engines' earlier frames are back as
    // well; with four slots that wait reached the submitters as back-pressure (plugin 78 - 91 k frames/s from run to run; six: 93 k beside the C ABI's 96 k)
    const int S = std::max(3, std::min(16, env_int("ZLY_STAGE_SLOTS", 6)));
    const size_t frame = (size_t)e->cfg.model_w * e->cfg.model_h * 3;
    size_t bytes = (size_t)((double)e->cfg.max_batch * (double)frame * 1.25);
    if (bytes < (8u << 20)) bytes = 8u << 20;
    if (env_int("ZLY_STAGE_MB", 0) > 0) bytes = (size_t)env_int("ZLY_STAGE_MB", 0) << 20;
    g->slot_bytes = (bytes + 4095) / 4096 * 4096;
    g->depth = std::max(1, std::min(S - 2, env_int("ZLY_INFLIGHT", 2)));
    g->slots.resize((size_t)S);
    const int dv = e->dev & 15;
    g->use_d2h_stream = env_int("ZLY_D2H_STREAM", 0) != 0;
    bool ok = (g_h2d_stream[dv] || hipStreamCreateWithFlags(&g_h2d_stream[dv], hipStreamNonBlocking) == hipSuccess) &&
              (!g->use_d2h_stream || g_d2h_stream[dv] || hipStreamCreateWithFlags(&g_d2h_stream[dv], hipStreamNonBlocking) == hipSuccess);
    g->copy_stream = g_h2d_stream[dv]; g->d2h_stream = g_d2h_stream[dv];
    const size_t sb = slab_bytes_of(e) * (size_t)e->cfg.max_batch;
    for (IngestSlot& sl : g->slots) {
        ok = ok && hipHostMalloc((void**)&sl.h_stage, g->slot_bytes, hipHostMallocDefault) == hipSuccess &&
             hipMalloc((void**)&sl.d_stage, g->slot_bytes) == hipSuccess &&
             hipHostMalloc((void**)&sl.h_slabs, sb, hipHostMallocDefault) == hipSuccess && hipMalloc((void**)&sl.d_slabs, sb) == hipSuccess &&
             hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming) == hipSuccess;
        if (ok) hipMemset(sl.d_slabs, 0, sb);
        sl.w.resize((size_t)e->cfg.max_batch); sl.h.resize((size_t)e->cfg.max_batch); sl.off.resize((size_t)e->cfg.max_batch);
        sl.consumed.resize((size_t)e->cfg.max_batch);
    }
    if (!ok) { ingest_free(e, g); return fail(ZLY_ERR_SYSTEM, "allocation of the staging ring failed"); }
    g->dispatcher = std::thread(ingest_dispatch_loop, e, g);
    g->completer = std::thread(ingest_complete_loop, e, g);
    e->ingest.store(g);
    return ZLY_OK;
}

static void ingest_destroy(zly_engine* e)
{
    Ingest* g = e->ingest.load();
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->stop = true;
    }
    g->cv_disp.notify_all(); g->cv_comp.notify_all(); g->cv_free.notify_all(); g->cv_done.notify_all();
    if (g->dispatcher.joinable()) g->dispatcher.join();
    if (g->completer.joinable()) g->completer.join();
    e->ingest.store(nullptr);
    if (g->copy_stream) hipStreamSynchronize(g->copy_stream);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->nms_stream) hipStreamSynchronize(e->nms_stream);
    if (g->d2h_stream) hipStreamSynchronize(g->d2h_stream);
    ExclusiveGate x;                                   // releases: alone in the process
    ingest_free(e, g);
}

// caller holds the exclusive gate and has drained the streams
static void ingest_free(zly_engine* e, Ingest* g)
{
    (void)e;
    for (IngestSlot& sl : g->slots) {
        if (sl.h_stage) hipHostFree(sl.h_stage);
        if (sl.d_stage) hipFree(sl.d_stage);
        if (sl.h_slabs) hipHostFree(sl.h_slabs);
        if (sl.d_slabs) hipFree(sl.d_slabs);
        if (sl.ev_h2d) hipEventDestroy(sl.ev_h2d);
        if (sl.ev_done) hipEventDestroy(sl.ev_done);
        if (sl.ev_out) hipEventDestroy(sl.ev_out);
    }
    delete g;                                          // the copy streams are shared by the process and stay
}

static int ingest_submit(zly_engine* e, const uint8_t* bgr, size_t nbytes, int32_t w, int32_t h, uint64_t* ticket, bool nonblock = false)
{
    if (!bgr || w <= 0 || h <= 0 || nbytes != (size_t)w * (size_t)h * 3u) {
        with_stats(e, [](zly_stats& st) { st.inference_errors++; });
        return fail(ZLY_ERR_INVALID_INPUT, "Invalid image data size: expected " + std::to_string((size_t)(w > 0 ? w : 0) * (size_t)(h > 0 ? h : 0) * 3u) +
                                               ", got " + std::to_string(nbytes));
    }
    if (!e->ingest.load()) {
        int rc = ingest_start(e);
        if (rc != ZLY_OK) return rc;
    }
    Ingest* g = e->ingest.load();
    const size_t padded = (nbytes + 15) / 16 * 16;
    if (padded > g->slot_bytes)
        return fail(ZLY_ERR_INVALID_INPUT, "frame of " + std::to_string(nbytes) + " bytes does not fit a staging slot of " + std::to_string(g->slot_bytes) + " bytes (ZLY_STAGE_MB)");
    IngestSlot* sl = nullptr;
    int idx = 0;
    size_t off = 0;
    {
        std::unique_lock<std::mutex> lk(g->mu);
        while (true) {
            if (g->stop) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
            if (g->open < 0 && !ingest_try_open(e, g)) {                                       // every ring slot is busy: back-pressure
                if (nonblock) return ZLY_PENDING;                                                // zly_submit_try: the caller offers the frame elsewhere
                g->cv_free.wait(lk);
                continue;
            }
            IngestSlot& o = g->slots[(size_t)g->open];
            if (o.n_reserved >= e->cfg.max_batch || o.bytes_used + padded > g->slot_bytes) {   // full: the dispatcher takes it from here
                ingest_close_open(g);
                g->cv_disp.notify_one();
                continue;
            }
            sl = &o;
            idx = o.n_reserved++;
            off = o.bytes_used;
            o.bytes_used += padded;
            o.w[(size_t)idx] = w; o.h[(size_t)idx] = h; o.off[(size_t)idx] = off;
            *ticket = (o.batch << 16) | (uint64_t)idx;
            if (o.n_reserved == e->cfg.max_batch) ingest_close_open(g);
            break;
        }
    }
    memcpy(sl->h_stage + off, bgr, nbytes);               // the one host copy of the request, on the caller's thread
    {
        std::lock_guard<std::mutex> lk(g->mu);
        sl->n_committed++;
    }
    g->cv_disp.notify_one();
    return ZLY_OK;
}

// under g->mu: the slot a ticket lives in, or null
static IngestSlot* ingest_slot_of(Ingest* g, uint64_t ticket, int* idx)
{
    const uint64_t batch = ticket >> 16;
    *idx = (int)(ticket & 0xffffu);
    IngestSlot& sl = g->slots[(size_t)(batch % g->slots.size())];
    if (sl.state == SLOT_FREE || sl.batch != batch || *idx >= sl.n_reserved || sl.consumed[(size_t)*idx]) return nullptr;   // consumed: 1 = done, 2 = claimed by a waiter
    return &sl;
}

static int ingest_poll(zly_engine* e, uint64_t ticket)
{
    Ingest* g = e->ingest.load();
    if (!g) return fail(ZLY_ERR_INVALID_ARGUMENT, "unknown ticket");
    std::lock_guard<std::mutex> lk(g->mu);
    int idx = 0;
    IngestSlot* sl = ingest_slot_of(g, ticket, &idx);
    if (!sl) return fail(ZLY_ERR_INVALID_ARGUMENT, "unknown or already consumed ticket");
    return sl->state == SLOT_DONE ? ZLY_OK : ZLY_PENDING;
}

static int ingest_wait(zly_engine* e, uint64_t ticket, zly_det* out, int32_t cap, int32_t* n_out)
{
    Ingest* g = e->ingest.load();
    if (!g) return fail(ZLY_ERR_INVALID_ARGUMENT, "unknown ticket");
    std::unique_lock<std::mutex> lk(g->mu);
    int idx = 0;
    IngestSlot* sl = ingest_slot_of(g, ticket, &idx);
    if (!sl) return fail(ZLY_ERR_INVALID_ARGUMENT, "unknown or already consumed ticket");
    // the ticket is CLAIMED before the wait: a second zly_wait on it (concurrent, or entered before this one finishes) fails instead of
    // consuming it twice -- two consumptions freed the ring slot while other tickets of the batch were still unread
    sl->consumed[(size_t)idx] = 2;
    g->cv_done.wait(lk, [&] { return g->stop || sl->state == SLOT_DONE; });
    if (sl->state != SLOT_DONE) { sl->consumed[(size_t)idx] = 0; return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running"); }
    const int rc = sl->rc;
    std::string err = sl->err;
    const uint64_t ts = sl->ts_ms;
    lk.unlock();
    int kept = 0;
    if (rc == ZLY_OK) {                                    // the slot cannot be recycled before this ticket is consumed below
        const size_t sb = slab_bytes_of(e);
        const zly_slab_header* hd = reinterpret_cast<const zly_slab_header*>(sl->h_slabs + sb * (size_t)idx);
        const zly_det* d = reinterpret_cast<const zly_det*>(hd + 1);
        kept = hd->n_kept;
        int m = std::min(std::min(kept, e->cfg.max_dets), (int)cap);
        for (int k = 0; k < m; ++k) { out[k] = d[k]; out[k].timestamp = ts; }
    }
    lk.lock();
    sl->consumed[(size_t)idx] = 1;
    if (++sl->n_consumed == sl->n_reserved) {
        sl->state = SLOT_FREE;
        g->cv_free.notify_all();
        g->cv_disp.notify_one();
    }
    lk.unlock();
    if (rc != ZLY_OK) return fail(rc, err);
    *n_out = kept;
    return ZLY_OK;
}

}  // namespace zly

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* zly_last_error(void) { return g_last_error.c_str(); }
const char* zly_version(void) { return "zly-hip 0.2 (gfx950)"; }

void zly_default_config(zly_config* c)
{
    memset(c, 0, sizeof *c);
    c->weights_path = nullptr;
    c->model_w = 416; c->model_h = 416;          // configs/server.json:30-31
    c->conf_thr = 0.5f; c->iou_thr = 0.45f;      // configs/server.json:7-8
    c->max_batch = 1; c->max_dets = 64; c->device = 0;
    c->dtype = ZLY_DTYPE_BF16; c->warmup_runs = 3; c->use_graph = 1;
}

// One untimed pass of the device-resident path at batch n on constant-128 frames, through run_path exactly as zly_detect_device and the
// pipelined host path run it (deferred NMS: both candidate-buffer parities): captures the batch-n graph(s) and replays each once, so that
// neither the capture (stream sync + capture + instantiate) nor a graph's first launch falls into a production call -- at start-up or
// after a hot reload.  warmupModel analogue for the throughput path (onnx_engine.cpp:919-954 warms its single-frame path only).
static int warm_batch(zly_engine* e, int n)
{
    std::lock_guard<std::mutex> lk(e->mu);
    const size_t fb = (size_t)e->cfg.model_w * e->cfg.model_h * 3;
    const size_t total = fb * (size_t)n;
    if (total > e->stage_bytes) {
        ExclusiveGate x;
        int rc = ensure_stage(e, total);
        if (rc != ZLY_OK) return rc;
    }
    SharedGate gl;
    HIP_TRY(hipMemsetAsync(e->d_stage, 128, total, e->stream), ZLY_ERR_INFERENCE);
    std::vector<int32_t> ws((size_t)n, e->cfg.model_w), hs((size_t)n, e->cfg.model_h);
    std::vector<size_t> offs((size_t)n);
    for (int i = 0; i < n; ++i) offs[(size_t)i] = (size_t)i * fb;
    int rc = set_desc(e, n, ws.data(), hs.data(), offs.data(), e->stream);
    if (rc != ZLY_OK) return rc;
    const bool defer = (e->cfg.flags & ZLY_FLAG_ASYNC_NMS) != 0;
    for (int k = 0; k < (defer ? 4 : 2); ++k) {
        rc = run_path(e, n, e->d_stage, nullptr, 0, e->stream, true, defer);
        if (rc != ZLY_OK) return rc;
    }
    gl.release();
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_INFERENCE);
    if (e->nms_stream) HIP_TRY(hipStreamSynchronize(e->nms_stream), ZLY_ERR_INFERENCE);
    return ZLY_OK;
}

int32_t zly_create(const zly_config* cfg, zly_engine** out)
{
    if (!cfg || !out) return fail(ZLY_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (cfg->model_w <= 0 || cfg->model_h <= 0 || cfg->model_w % 32 || cfg->model_h % 32)
        return fail(ZLY_ERR_INVALID_ARGUMENT, "model_w/model_h must be positive multiples of 32");
    if (cfg->max_batch < 1 || cfg->max_dets < 1) return fail(ZLY_ERR_INVALID_ARGUMENT, "max_batch/max_dets must be >= 1");
    if (cfg->max_batch > 65535) return fail(ZLY_ERR_INVALID_ARGUMENT, "max_batch must be <= 65535 (a ticket of the pipelined path carries the frame's index in 16 bits)");
    if (cfg->dtype != ZLY_DTYPE_BF16 && cfg->dtype != ZLY_DTYPE_FP32) return fail(ZLY_ERR_INVALID_ARGUMENT, "dtype must be ZLY_DTYPE_FP32 or ZLY_DTYPE_BF16");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(ZLY_ERR_SYSTEM, "no HIP device available: this engine has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(ZLY_ERR_INVALID_ARGUMENT, "device ordinal out of range");

    zly_engine* e = new zly_engine();
    e->cfg = *cfg;
    e->weights_path = cfg->weights_path ? cfg->weights_path : "";
    e->cfg.weights_path = nullptr;
    e->dev = cfg->device;
    e->dtype = cfg->dtype;
    e->esz = cfg->dtype == ZLY_DTYPE_BF16 ? 2 : 4;
    e->sw.no_c2f = getenv("ZLY_NO_C2F") != nullptr;
#ifdef ZLY_DIAG
    if (const char* v = getenv("ZLY_ABLATE_SKIP")) e->sw.ablate = std::string(",") + v + ",";       // result-changing: compiled into libzly_diag.so only (tools/ablate_launches.sh)
#endif
    e->sw.no_det_merge = getenv("ZLY_NO_DET_MERGE") != nullptr;
    e->sw.no_tail_split = getenv("ZLY_NO_TAIL_SPLIT") != nullptr;
    e->sw.no_lanes = getenv("ZLY_NO_LANES") != nullptr || getenv("ZLY_CU_PART") != nullptr;
    e->sw.no_sppf = getenv("ZLY_SPPF_FUSED") == nullptr;                         // the fused SPPF kernel is OPT-IN (ZLY_SPPF_FUSED=1): parity-green, 36 -> ~24 us in isolation at batch 64, but the step gets 0.5 % slower
    e->sw.pool_six_pass = getenv("ZLY_SPPF_POOL_LDS") != nullptr;                 // tuning / tests: SPPF's pools on the six-pass LDS kernel also on small maps
    e->sw.no_wsk = getenv("ZLY_NO_WSK") != nullptr;                               // tuning / tests: the class-branch convs on the LDS-tiled kernel (96-channel padding)
    e->sw.nms_general = getenv("ZLY_NMS_GENERAL") != nullptr;                     // tests / A-B: every frame on NMS's eight-wave path
    if (const char* v = getenv("ZLY_STEM1_NW")) e->sw.stem1_nw = atoi(v);           // tuning aids: waves per workgroup of the front kernel (12 / 16), ...
    if (const char* v = getenv("ZLY_STEM1_GRID")) e->sw.stem1_grid = atoi(v);       // ... workgroups of its persistent grid ...
    if (const char* v = getenv("ZLY_STEM1_VAR")) e->sw.stem1_var = atoi(v);         // ... and 0 = round 3's staging / tap order (A/B on one box)
    std::string err;
    int rc = load_zlyw(e->weights_path.c_str(), &e->model, &err);        // host only: file parse
    if (rc != ZLY_OK) { delete e; return fail(rc, err); }
    e->nc = e->model.nc;
    PlanState plan_state;
    rc = build_plan_host(e, &plan_state, &err);                          // host only: op list + weight repack, outside the gate
    if (rc != ZLY_OK) { delete e; return fail(rc, err); }

    // Everything that allocates on / uploads to the device runs alone in the process (ExclusiveGate): a hot reload builds new engines beside
    // running ones, and a hipMalloc / hipMemcpy here while one of them had a stream capture open failed both sides.
    auto device_setup = [&]() -> int {
        HIP_TRY(hipSetDevice(cfg->device), ZLY_ERR_SYSTEM);
        HIP_TRY(conv_init(), ZLY_ERR_SYSTEM);
        HIP_TRY(pair_init(), ZLY_ERR_SYSTEM);
        HIP_TRY(stem1_init(), ZLY_ERR_SYSTEM);
        HIP_TRY(c2f_init(), ZLY_ERR_SYSTEM);
        HIP_TRY(sppf_init(), ZLY_ERR_SYSTEM);
        HIP_TRY(nms_init(), ZLY_ERR_SYSTEM);
        // ZLY_CU_PART="i/n": this engine's streams only use the i-th of n equal slices of the chip's compute units (spatial
        // partitioning: several engines run side by side, the launch-latency-bound small-map layers of one beside the
        // bandwidth-bound layers of another).  Experiment switch: see DESIGN.md section 5.
        if (const char* cp = getenv("ZLY_CU_PART")) {
            int pi = 0, pn = 1;
            if (sscanf(cp, "%d/%d", &pi, &pn) == 2 && pn >= 1 && pn <= 8 && pi >= 0 && pi < pn) {
                e->cu_part_n = pn;
                const int per = 256 / pn;
                for (int b = pi * per; b < (pi + 1) * per; ++b) e->cu_mask[b / 32] |= 1u << (b % 32);
                set_num_cus(per);
            }
        }
        auto make_stream = [&](hipStream_t* st) {
            if (e->cu_part_n > 1) return hipExtStreamCreateWithCUMask(st, 8, e->cu_mask) == hipSuccess;
            return hipStreamCreateWithFlags(st, hipStreamNonBlocking) == hipSuccess;
        };
        bool sok = make_stream(&e->stream);
        // side streams only when the Detect-branch lanes can be used: every stream beyond the hardware-queue limit shares a hardware queue
        // with another one, and two streams on one queue serialise -- several engines per GPU want few streams each
        const bool want_lanes = !(cfg->flags & ZLY_FLAG_SINGLE_CHAIN) && !e->sw.no_lanes;
        for (int i = 0; i < 2 && sok && want_lanes; ++i) {
            sok = make_stream(&e->side[i]) &&
                  hipEventCreateWithFlags(&e->ev_fork[i], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming) == hipSuccess;
        }
        if (!sok) return fail(ZLY_ERR_SYSTEM, "hipStreamCreate failed");
        int prc = build_plan_device(e, &plan_state, &err);
        if (prc != ZLY_OK) return fail(prc, err);
        plan_state.blob = std::vector<uint8_t>();      // the host image of the weight blob is on the device now

        const size_t B = (size_t)cfg->max_batch, N = (size_t)e->N;
        bool ok = true;
        ok = ok && hipMalloc((void**)&e->d_head, ((cfg->flags & ZLY_FLAG_NO_HEAD_TENSOR) ? 1 : B) * (4 + (size_t)e->nc) * N * sizeof(float)) == hipSuccess;
        ok = ok && hipMalloc((void**)&e->d_cand, B * N * sizeof(Cand)) == hipSuccess;
        ok = ok && hipMalloc((void**)&e->d_scratch, B * N * sizeof(Cand)) == hipSuccess;
        ok = ok && hipMalloc((void**)&e->d_count, B * sizeof(int)) == hipSuccess;
        ok = ok && hipMalloc((void**)&e->d_slabs, B * slab_bytes_of(e)) == hipSuccess;
        ok = ok && hipMalloc((void**)&e->d_desc, B * sizeof(FrameDesc)) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&e->h_desc, zly_engine::DESC_RING * B * sizeof(FrameDesc), hipHostMallocDefault) == hipSuccess;
        for (int i = 0; i < zly_engine::DESC_RING && ok; ++i) ok = hipEventCreateWithFlags(&e->ev_desc[i], hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 4 && ok; ++i) ok = hipEventCreate(&e->ev_t[i]) == hipSuccess;
        for (int i = 0; i < 2 && ok; ++i) ok = hipEventCreateWithFlags(&e->ev_call[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&e->h_slabs, B * slab_bytes_of(e), hipHostMallocDefault) == hipSuccess;
        if (!ok) return fail(ZLY_ERR_SYSTEM, "device allocation failed");
        ok = hipMemset(e->d_slabs, 0, B * slab_bytes_of(e)) == hipSuccess && hipMemset(e->d_count, 0, B * sizeof(int)) == hipSuccess;
        e->cur_cand = e->d_cand; e->cur_count = e->d_count;
        if (ok && (cfg->flags & ZLY_FLAG_ASYNC_NMS)) {
            ok = hipMalloc((void**)&e->d_cand_alt, B * N * sizeof(Cand)) == hipSuccess && hipMalloc((void**)&e->d_count_alt, B * sizeof(int)) == hipSuccess &&
                 make_stream(&e->nms_stream) &&
                 hipEventCreateWithFlags(&e->ev_head, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&e->ev_nms[0], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&e->ev_nms[1], hipEventDisableTiming) == hipSuccess &&
                 hipMemset(e->d_count_alt, 0, B * sizeof(int)) == hipSuccess;
            if (!ok) return fail(ZLY_ERR_SYSTEM, "device allocation failed (deferred NMS)");
        }
        if (!ok) return fail(ZLY_ERR_SYSTEM, "device initialisation failed");
        // staging for one frame (zly_detect) -- or, with a warm-up, for max_batch model-sized frames (warm_batch; zly_detect_batch then never allocates)
        const size_t fb = (size_t)cfg->model_w * cfg->model_h * 3;
        return ensure_stage(e, cfg->warmup_runs > 0 ? fb * B : fb);
    };
    {
        ExclusiveGate x;
        rc = device_setup();
    }
    if (rc != ZLY_OK) { std::string m = g_last_error; destroy_engine(e); return fail(rc, m); }

    // warmupModel analogue (onnx_engine.cpp:919-954): constant-128 frames of model size through the single-frame path, then the
    // throughput path at max_batch.  Both go through the normal entry points' machinery (gate, capture, replay).
    if (cfg->warmup_runs > 0) {
        const size_t fb = (size_t)cfg->model_w * cfg->model_h * 3;
        std::vector<uint8_t> grey(fb, 128);
        std::vector<zly_det> dets((size_t)cfg->max_dets);
        for (int i = 0; i < cfg->warmup_runs && rc == ZLY_OK; ++i) {
            int32_t nd = 0;
            rc = zly_detect(e, grey.data(), fb, cfg->model_w, cfg->model_h, dets.data(), cfg->max_dets, &nd);
        }
        if (rc == ZLY_OK && cfg->max_batch > 1) rc = warm_batch(e, cfg->max_batch);
        if (rc != ZLY_OK) { std::string m = g_last_error; destroy_engine(e); return fail(rc, "warm-up failed: " + m); }
        with_stats(e, [](zly_stats& st) { st = zly_stats{}; });
        e->sample_ctr = 0; e->t_pending = false;
    }
    *out = e;
    return ZLY_OK;
}

int32_t zly_destroy(zly_engine* e)
{
    if (!e) return ZLY_OK;
    const hipError_t r = destroy_engine(e);
    if (r != hipSuccess) return fail(ZLY_ERR_SYSTEM, std::string("zly_destroy: ") + hipGetErrorString(r));
    return ZLY_OK;
}

static int detect_host_locked(zly_engine* e, int32_t n, const uint8_t* const* bgr, const size_t* nbytes,
                              const int32_t* w, const int32_t* h, zly_det* out, int32_t cap, int32_t* n_out)
{
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    std::vector<size_t> offs((size_t)n);
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (!bgr[i] || w[i] <= 0 || h[i] <= 0 || nbytes[i] != (size_t)w[i] * (size_t)h[i] * 3u) {
            with_stats(e, [](zly_stats& st) { st.inference_errors++; });
            return fail(ZLY_ERR_INVALID_INPUT, "Invalid image data size: expected " + std::to_string((size_t)(w[i] > 0 ? w[i] : 0) * (size_t)(h[i] > 0 ? h[i] : 0) * 3u) +
                                                   ", got " + std::to_string(nbytes[i]));
        }
        offs[(size_t)i] = total;
        total += (nbytes[i] + 15) / 16 * 16;
    }
    int rc = ZLY_OK;
    if (total > e->stage_bytes) {                                        // growth beyond what zly_create staged (frames larger than the model input): an allocation, alone in the process
        ExclusiveGate x;
        rc = ensure_stage(e, total);
        if (rc != ZLY_OK) return rc;
    }
    // waiting for the device and the host copy are not enqueue sections: no gate is held across them (other engines keep enqueuing)
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_INFERENCE);        // pinned staging is reused call to call
    for (int i = 0; i < n; ++i) memcpy(e->h_stage + offs[(size_t)i], bgr[i], nbytes[i]);
    SharedGate gl;
    HIP_TRY(hipMemcpyAsync(e->d_stage, e->h_stage, total, hipMemcpyHostToDevice, e->stream), ZLY_ERR_INFERENCE);
    rc = set_desc(e, n, w, h, offs.data(), e->stream);
    if (rc != ZLY_OK) return rc;
    rc = run_path(e, n, e->d_stage, nullptr, 0, e->stream, true);
    if (rc != ZLY_OK) { with_stats(e, [](zly_stats& st) { st.inference_errors++; }); return rc; }
    const size_t sb = slab_bytes_of(e);
    HIP_TRY(hipMemcpyAsync(e->h_slabs, e->d_slabs, sb * (size_t)n, hipMemcpyDeviceToHost, e->stream), ZLY_ERR_INFERENCE);
    gl.release();                                                        // waiting for the device is not an enqueue section
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_INFERENCE);
    const uint64_t ts = now_ms();                                        // onnx_engine.cpp:813-815
    for (int i = 0; i < n; ++i) {
        const zly_slab_header* hd = reinterpret_cast<const zly_slab_header*>(e->h_slabs + sb * (size_t)i);
        const zly_det* d = reinterpret_cast<const zly_det*>(hd + 1);
        int m = hd->n_kept;
        if (m > e->cfg.max_dets) m = e->cfg.max_dets;
        if (m > cap) m = cap;
        for (int k = 0; k < m; ++k) { out[(size_t)i * cap + k] = d[k]; out[(size_t)i * cap + k].timestamp = ts; }
        n_out[i] = hd->n_kept;
    }
    harvest_timing(e);                                                   // the stream is idle: a sampled call's events are complete
    with_stats(e, [&](zly_stats& st) { st.inference_count += (uint64_t)n; });
    return ZLY_OK;
}

int32_t zly_detect(zly_engine* e, const uint8_t* bgr, size_t nbytes, int32_t w, int32_t h, zly_det* out, int32_t cap, int32_t* n_out)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!out || !n_out || cap < 0) return fail(ZLY_ERR_INVALID_ARGUMENT, "null output");
    std::lock_guard<std::mutex> lk(e->mu);
    const auto t0 = std::chrono::steady_clock::now();
    const uint8_t* ptrs[1] = {bgr};
    int rc = detect_host_locked(e, 1, ptrs, &nbytes, &w, &h, out, cap, n_out);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    with_stats(e, [&](zly_stats& st) { st.last_detect_ms = ms; });
    return rc;
}

int32_t zly_detect_batch(zly_engine* e, int32_t n, const uint8_t* const* bgr, const size_t* nbytes,
                         const int32_t* w, const int32_t* h, zly_det* out, int32_t cap, int32_t* n_out)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!bgr || !nbytes || !w || !h || !out || !n_out || cap < 0) return fail(ZLY_ERR_INVALID_ARGUMENT, "null argument");
    if (n < 1 || n > e->cfg.max_batch) return fail(ZLY_ERR_INVALID_ARGUMENT, "batch size out of range");
    std::lock_guard<std::mutex> lk(e->mu);
    return detect_host_locked(e, n, bgr, nbytes, w, h, out, cap, n_out);
}

int32_t zly_submit(zly_engine* e, const uint8_t* bgr, size_t nbytes, int32_t w, int32_t h, uint64_t* ticket)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!ticket) return fail(ZLY_ERR_INVALID_ARGUMENT, "null ticket");
    return ingest_submit(e, bgr, nbytes, w, h, ticket);
}

int32_t zly_submit_try(zly_engine* e, const uint8_t* bgr, size_t nbytes, int32_t w, int32_t h, uint64_t* ticket)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!ticket) return fail(ZLY_ERR_INVALID_ARGUMENT, "null ticket");
    return ingest_submit(e, bgr, nbytes, w, h, ticket, true);
}

int32_t zly_poll(zly_engine* e, uint64_t ticket)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    return ingest_poll(e, ticket);
}

int32_t zly_wait(zly_engine* e, uint64_t ticket, zly_det* out, int32_t cap, int32_t* n_out)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!out || !n_out || cap < 0) return fail(ZLY_ERR_INVALID_ARGUMENT, "null output");
    return ingest_wait(e, ticket, out, cap, n_out);
}

int32_t zly_detect_device(zly_engine* e, int32_t n, const void* d_frames, int32_t w, int32_t h, void* d_slabs, uint32_t frame_tag0, void* stream)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!d_frames || w <= 0 || h <= 0) return fail(ZLY_ERR_INVALID_INPUT, "bad frame pointer or size");
    if (n < 1 || n > e->cfg.max_batch) return fail(ZLY_ERR_INVALID_ARGUMENT, "batch size out of range");
    std::lock_guard<std::mutex> lk(e->mu);
    SharedGate gl;
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    std::vector<int32_t> ws((size_t)n, w), hs((size_t)n, h);
    std::vector<size_t> offs((size_t)n);
    for (int i = 0; i < n; ++i) offs[(size_t)i] = (size_t)i * (size_t)w * (size_t)h * 3u;
    int rc = set_desc(e, n, ws.data(), hs.data(), offs.data(), s);
    if (rc != ZLY_OK) return rc;
    hipStream_t ns = s;
    rc = run_path(e, n, (const uint8_t*)d_frames, d_slabs, frame_tag0, s, true, (e->cfg.flags & ZLY_FLAG_ASYNC_NMS) != 0, &ns);
    if (rc != ZLY_OK) { with_stats(e, [](zly_stats& st) { st.inference_errors++; }); return rc; }
    HIP_TRY(hipEventRecord(e->ev_call[e->call_seq & 1], ns), ZLY_ERR_INFERENCE);
    e->call_seq++;
    with_stats(e, [&](zly_stats& st) { st.inference_count += (uint64_t)n; });
    return ZLY_OK;
}

int32_t zly_join(zly_engine* e, void* stream, int32_t lag)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    // completion events are kept for the last two calls only: a larger lag could not be ordered and used to return ZLY_OK without ordering anything
    if (lag < 0 || lag > 1) return fail(ZLY_ERR_INVALID_ARGUMENT, "lag must be 0 or 1");
    std::lock_guard<std::mutex> lk(e->mu);
    SharedGate gl;
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    // every call but the last `lag`: calls are stream-ordered among themselves, so waiting for call (last - lag) covers all earlier ones
    if (e->call_seq > (uint64_t)lag)
        HIP_TRY(hipStreamWaitEvent(s, e->ev_call[(e->call_seq - 1 - (uint64_t)lag) & 1], 0), ZLY_ERR_INFERENCE);
    return join_nms(e, s, lag);
}

size_t zly_slab_bytes(const zly_engine* e) { return e ? slab_bytes_of(e) : 0; }

int32_t zly_sync(zly_engine* e)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_INFERENCE);
    if (e->nms_stream) HIP_TRY(hipStreamSynchronize(e->nms_stream), ZLY_ERR_INFERENCE);
    harvest_timing(e);
    return ZLY_OK;
}

int32_t zly_read_slabs(zly_engine* e, int32_t n, void* host_slabs)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!host_slabs || n < 1 || n > e->cfg.max_batch) return fail(ZLY_ERR_INVALID_ARGUMENT, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    {
        SharedGate gl;
        int rcj = join_nms(e, e->stream);
        if (rcj != ZLY_OK) return rcj;
        HIP_TRY(hipMemcpyAsync(host_slabs, e->d_slabs, slab_bytes_of(e) * (size_t)n, hipMemcpyDeviceToHost, e->stream), ZLY_ERR_INFERENCE);
    }
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_INFERENCE);
    return ZLY_OK;
}

int32_t zly_preprocess(zly_engine* e, const uint8_t* bgr, size_t nbytes, int32_t w, int32_t h, float* out_nchw)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!out_nchw) return fail(ZLY_ERR_INVALID_ARGUMENT, "null output");
    if (!bgr || w <= 0 || h <= 0 || nbytes != (size_t)w * (size_t)h * 3u)
        return fail(ZLY_ERR_INVALID_INPUT, "Invalid image data size: expected " + std::to_string((size_t)(w > 0 ? w : 0) * (size_t)(h > 0 ? h : 0) * 3u) + ", got " + std::to_string(nbytes));
    std::lock_guard<std::mutex> lk(e->mu);
    ExclusiveGate gl;                                    // parity / debug entry point: allocations and copies, alone in the process
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    int rc = ensure_stage(e, nbytes);
    if (rc != ZLY_OK) return rc;
    const size_t elems = (size_t)3 * e->cfg.model_w * e->cfg.model_h;
    rc = ensure_scratch_f32(e, elems);
    if (rc != ZLY_OK) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_INFERENCE);
    memcpy(e->h_stage, bgr, nbytes);
    HIP_TRY(hipMemcpyAsync(e->d_stage, e->h_stage, nbytes, hipMemcpyHostToDevice, e->stream), ZLY_ERR_INFERENCE);
    size_t off0 = 0;
    rc = set_desc(e, 1, &w, &h, &off0, e->stream);
    if (rc != ZLY_OK) return rc;
    HIP_TRY(launch_preprocess(e->dtype, e->d_stage, e->d_desc, 1, nullptr, e->d_scratch_f32, e->cfg.model_w, e->cfg.model_h, e->stream), ZLY_ERR_INFERENCE);
    HIP_TRY(hipMemcpyAsync(out_nchw, e->d_scratch_f32, elems * sizeof(float), hipMemcpyDeviceToHost, e->stream), ZLY_ERR_INFERENCE);
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_INFERENCE);
    return ZLY_OK;
}

int32_t zly_forward(zly_engine* e, int32_t n, const float* images_nchw, float* head_out)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!images_nchw || !head_out) return fail(ZLY_ERR_INVALID_ARGUMENT, "null argument");
    if (n < 1 || n > e->cfg.max_batch) return fail(ZLY_ERR_INVALID_ARGUMENT, "batch size out of range");
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    const size_t elems = (size_t)n * 3 * e->cfg.model_w * e->cfg.model_h;
    int rc = ZLY_OK;
    {
        ExclusiveGate x;                                      // scratch allocation: alone in the process
        rc = ensure_scratch_f32(e, elems);
    }
    if (rc != ZLY_OK) return rc;
    SharedGate gl;
    HIP_TRY(hipMemcpyAsync(e->d_scratch_f32, images_nchw, elems * sizeof(float), hipMemcpyHostToDevice, e->stream), ZLY_ERR_INFERENCE);
    HIP_TRY(launch_nchw_to_nhwc8(e->dtype, e->d_scratch_f32, e->bufs[(size_t)e->in_buf].ptr, n, e->cfg.model_w, e->cfg.model_h, e->stream), ZLY_ERR_INFERENCE);
    // frames are model-sized for the purposes of the (unused) decode that follows
    std::vector<int32_t> ws((size_t)n, e->cfg.model_w), hs((size_t)n, e->cfg.model_h);
    std::vector<size_t> offs((size_t)n, 0);
    rc = set_desc(e, n, ws.data(), hs.data(), offs.data(), e->stream);
    if (rc != ZLY_OK) return rc;
    if (e->cfg.flags & ZLY_FLAG_NO_HEAD_TENSOR) return fail(ZLY_ERR_INVALID_ARGUMENT, "engine was created with ZLY_FLAG_NO_HEAD_TENSOR: the head tensor is not materialised");
    rc = run_path(e, n, nullptr, nullptr, 0, e->stream, false);
    if (rc != ZLY_OK) return rc;
    HIP_TRY(hipMemcpyAsync(head_out, e->d_head, (size_t)n * (4 + (size_t)e->nc) * e->N * sizeof(float), hipMemcpyDeviceToHost, e->stream), ZLY_ERR_INFERENCE);
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_INFERENCE);
    return ZLY_OK;
}

int32_t zly_head_tensor(zly_engine* e, int32_t idx, float* head_out)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!head_out || idx < 0 || idx >= e->cfg.max_batch) return fail(ZLY_ERR_INVALID_ARGUMENT, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    SharedGate gl;
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    if (e->cfg.flags & ZLY_FLAG_NO_HEAD_TENSOR) return fail(ZLY_ERR_INVALID_ARGUMENT, "engine was created with ZLY_FLAG_NO_HEAD_TENSOR: the head tensor is not materialised");
    const size_t per = (4 + (size_t)e->nc) * e->N;
    HIP_TRY(hipMemcpyAsync(head_out, e->d_head + per * (size_t)idx, per * sizeof(float), hipMemcpyDeviceToHost, e->stream), ZLY_ERR_INFERENCE);
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_INFERENCE);
    return ZLY_OK;
}

int32_t zly_postprocess(zly_engine* e, const float* head, int32_t num_classes, int32_t num_boxes, int32_t img_w, int32_t img_h,
                        float conf_thr, float iou_thr, zly_det* out, int32_t cap, int32_t* n_out, int32_t* n_candidates)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!head || !out || !n_out || num_classes < 1 || num_classes > 1024 || num_boxes < 0 || cap < 1 || img_w <= 0 || img_h <= 0)
        return fail(ZLY_ERR_INVALID_ARGUMENT, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    ExclusiveGate gl;                                    // parity / debug entry point: allocations and copies, alone in the process
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    *n_out = 0;
    if (n_candidates) *n_candidates = 0;
    if (num_boxes == 0) return ZLY_OK;
    const size_t N = (size_t)num_boxes;
    const size_t head_bytes = (4 + (size_t)num_classes) * N * sizeof(float);
    const size_t slab = sizeof(zly_slab_header) + (size_t)cap * sizeof(zly_det);
    // one-off device scratch: [head | cand | scratch | count | desc | slab]
    size_t off_cand = (head_bytes + 255) / 256 * 256;
    size_t off_scr = off_cand + (N * sizeof(Cand) + 255) / 256 * 256;
    size_t off_cnt = off_scr + (N * sizeof(Cand) + 255) / 256 * 256;
    size_t off_desc = off_cnt + 256;
    size_t off_slab = off_desc + 256;
    size_t total = off_slab + slab;
    char* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, total), ZLY_ERR_SYSTEM);
    std::vector<unsigned char> hslab(slab);
    FrameDesc fd; fd.src_off = 0; fd.w = img_w; fd.h = img_h;
    hipError_t r = hipMemcpyAsync(d, head, head_bytes, hipMemcpyHostToDevice, e->stream);
    if (r == hipSuccess) r = hipMemcpyAsync(d + off_desc, &fd, sizeof fd, hipMemcpyHostToDevice, e->stream);
    if (r == hipSuccess) r = hipMemsetAsync(d + off_cnt, 0, sizeof(int), e->stream);
    if (r == hipSuccess) r = launch_decode((const float*)d, num_classes, num_boxes, 1, (const FrameDesc*)(d + off_desc), conf_thr,
                                           (Cand*)(d + off_cand), (int*)(d + off_cnt), e->stream);
    if (r == hipSuccess) r = launch_nms((const Cand*)(d + off_cand), (int*)(d + off_cnt), num_boxes, 1, iou_thr, num_classes,
                                        (Cand*)(d + off_scr), d + off_slab, cap, 0, e->stream, e->sw.nms_general ? 1 : 0);
    if (r == hipSuccess) r = hipMemcpyAsync(hslab.data(), d + off_slab, slab, hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    hipFree(d);
    if (r != hipSuccess) return fail(ZLY_ERR_INFERENCE, std::string("postprocess: ") + hipGetErrorString(r));
    const zly_slab_header* hd = reinterpret_cast<const zly_slab_header*>(hslab.data());
    const zly_det* dd = reinterpret_cast<const zly_det*>(hd + 1);
    const int m = hd->n_kept < cap ? hd->n_kept : cap;
    for (int k = 0; k < m; ++k) out[k] = dd[k];
    *n_out = hd->n_kept;
    if (n_candidates) *n_candidates = hd->n_candidates;
    return ZLY_OK;
}

int32_t zly_debug_tap(zly_engine* e, const char* name, int32_t idx, float* out, size_t cap_floats, int32_t* c, int32_t* h, int32_t* w)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!name || !out || idx < 0 || idx >= e->cfg.max_batch) return fail(ZLY_ERR_INVALID_ARGUMENT, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    ExclusiveGate gl;                                    // parity / debug entry point: allocations and copies, alone in the process
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    int buf = -1, co = 0, C = 0;
    bool f32 = false;
    if (std::string(name) == "images") { buf = e->in_buf; co = 0; C = 3; }
    else {
        auto it = e->tap_index.find(name);
        auto jt = e->tap_final.find(name);
        if (it != e->tap_index.end()) {
            const Op& op = e->ops[(size_t)it->second.first];
            if (e->last_stem1 && op.name == "model.0" && !(e->cfg.flags & ZLY_FLAG_DUMP_LOGITS))
                return fail(ZLY_ERR_INVALID_ARGUMENT, "tap model.0 stays in LDS inside the fused stem kernel; create the engine with ZLY_FLAG_DUMP_LOGITS (or ZLY_FLAG_NO_FUSION)");
            if (e->last_n > 0 && op.c2f_leader >= 0 && c2f_active(e, e->ops[(size_t)op.c2f_leader], e->last_n) &&
                (op.c2f_vis == 2 || (op.c2f_vis == 1 && !(e->cfg.flags & ZLY_FLAG_DUMP_LOGITS))))
                return fail(ZLY_ERR_INVALID_ARGUMENT, std::string("tap ") + name + " stays in LDS inside the fused C2f kernel at this batch size; create the engine with " +
                                                          (op.c2f_vis == 1 ? "ZLY_FLAG_DUMP_LOGITS or " : "") + "ZLY_FLAG_NO_FUSION");
            if (it->second.first == e->sppf_cv1 && sppf_active(e) && !(e->cfg.flags & ZLY_FLAG_DUMP_LOGITS))
                return fail(ZLY_ERR_INVALID_ARGUMENT, std::string("tap ") + name + " stays in LDS inside the fused SPPF kernel; create the engine with ZLY_FLAG_DUMP_LOGITS or ZLY_FLAG_NO_FUSION");
            if (op.pair == 1 && e->last_n > 0 && pair_active(e, op, e->last_n))
                return fail(ZLY_ERR_INVALID_ARGUMENT, std::string("tap ") + name + " stays in LDS inside the fused bottleneck kernel at this batch size; create the engine with ZLY_FLAG_NO_FUSION");
            buf = op.out.buf; co = op.out.co + op.tap_co[(size_t)it->second.second]; C = op.tap_c[(size_t)it->second.second];
            f32 = op.out_f32 != 0;
        } else if (jt != e->tap_final.end()) {
            if (!(e->cfg.flags & ZLY_FLAG_DUMP_LOGITS))
                return fail(ZLY_ERR_INVALID_ARGUMENT, std::string("tap ") + name + " is computed inside the fused Detect kernel; create the engine with ZLY_FLAG_DUMP_LOGITS");
            buf = jt->second.first; co = jt->second.second; C = co == 0 ? 64 : e->nc; f32 = true;
        } else {
            return fail(ZLY_ERR_INVALID_ARGUMENT, std::string("unknown tap: ") + name);
        }
    }
    const Buffer& b = e->bufs[(size_t)buf];
    const size_t elems = (size_t)C * b.H * b.W;
    if (elems > cap_floats) return fail(ZLY_ERR_INVALID_ARGUMENT, "tap output buffer too small");
    int rc = ensure_scratch_f32(e, elems);
    if (rc != ZLY_OK) return rc;
    HIP_TRY(launch_tap_to_nchw(f32 ? ZLY_DTYPE_FP32 : e->dtype, b.ptr, b.C, co, C, b.H, b.W, idx, e->d_scratch_f32, e->stream), ZLY_ERR_INFERENCE);
    HIP_TRY(hipMemcpyAsync(out, e->d_scratch_f32, elems * sizeof(float), hipMemcpyDeviceToHost, e->stream), ZLY_ERR_INFERENCE);
    HIP_TRY(hipStreamSynchronize(e->stream), ZLY_ERR_INFERENCE);
    if (c) *c = C;
    if (h) *h = b.H;
    if (w) *w = b.W;
    return ZLY_OK;
}

int32_t zly_num_classes(const zly_engine* e) { return e ? e->nc : 0; }
int32_t zly_weights_fp8(const zly_engine* e) { return (e && e->model.fp8_weights) ? 1 : 0; }
int32_t zly_num_anchors(const zly_engine* e) { return e ? e->N : 0; }
int32_t zly_num_ops(const zly_engine* e) { return e ? (int32_t)e->ops.size() : 0; }

int32_t zly_op_info_at(const zly_engine* e, int32_t i, zly_op_info* out)
{
    if (!e || !out || i < 0 || i >= (int32_t)e->ops.size()) return fail(ZLY_ERR_INVALID_ARGUMENT, "bad argument");
    const Op& op = e->ops[(size_t)i];
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", op.name.c_str());
    out->kind = op.kind;
    out->flops_per_frame = op.flops;
    out->bytes_per_frame = op.bytes;
    return ZLY_OK;
}

int32_t zly_op_kernel_name(zly_engine* e, int32_t i, int32_t n, char* out, size_t cap)
{
    if (!e || !out || cap < 8 || i < 0 || i >= (int32_t)e->ops.size() || n < 1 || n > e->cfg.max_batch) return fail(ZLY_ERR_INVALID_ARGUMENT, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    const Op& op = e->ops[(size_t)i];
    std::string k;
    switch (op.kind) {
    case OP_PREPROCESS: k = e->stem_fused ? "(fused into stem_fused_kernel)" : "preprocess_kernel"; break;
    case OP_SPPF: {
        const Buffer& pb = e->bufs[(size_t)op.in.buf];
        k = sppf_active(e) ? "(fused into the SPPF kernel at model.9.cv1)"
          : (e->dtype == ZLY_DTYPE_BF16 && pb.H <= 16 && pb.W <= 16 && !e->sw.pool_six_pass && n <= 16) ? "sppf_pool16_kernel<DPP row windows, one barrier>" : "sppf_pool_kernel";
        break;
    }
    case OP_HEAD: k = op_is_noop(e, op, n) ? "(covered by the last tail launch)" : "head_fused_kernel"; break;
    case OP_NMS: k = "nms_kernel"; break;
    case OP_CONV: {
        if (i == 1 && e->stem1) { k = "stem_model1_kernel (preprocess+model.0+model.1)"; break; }
        if (i == 2 && e->stem1) { k = "(fused into the previous launch)"; break; }
        if (i == 1 && e->stem_fused) { k = "stem_fused_kernel"; break; }
        if (sppf_active(e) && (i == e->sppf_cv1 || i == e->sppf_cv2)) {
            k = i == e->sppf_cv2 ? std::string("(fused into the SPPF kernel at model.9.cv1)")
                                 : "sppf_fused_kernel<cv1+3 pools+cv2,SPLIT=" + std::to_string(sppf_split(e->ops[(size_t)e->sppf_cv2].cout, n)) + ">";
            break;
        }
        if (detect_merge_active(e, n) && detect_merge_role(e, i)) {
            const int role = detect_merge_role(e, i);
            k = role == 1 ? "(in a merged Detect launch)" : role == 2 ? "conv_igemm_multi_kernel<CT=3> (the 3 Detect stems)" : "conv_igemm_multi_kernel<CT=2> (the 6 Detect branch convs)";
            break;
        }
        if (c2f_covered(e, op, n)) { k = "(fused into the C2f kernel at " + op.c2f_leader_name + ")"; break; }
        if (c2f_active(e, op, n)) {
            const C2fPlan* pl = c2f_active(e, op, n);
            k = "c2f_kernel<C=" + std::to_string(op.c2f_c) + (op.c2f_c == 32 ? ",NW=" + std::to_string(pl->nw) : std::string()) +
                (op.c2f_mode == 3 ? ",cv1+bottleneck+cv2>" : op.c2f_mode == 1 ? ",cv1+bottleneck>" : ",bottleneck+cv2>");
            break;
        }
        if (op.pair && pair_active(e, op, n)) { k = op.pair == 1 ? "bottleneck_pair_kernel<" + std::to_string(op.pair_c) + ">" : "(fused into the previous launch)"; break; }
        if (wsk_active(e, op, n)) { k = "conv3x3_wsk_kernel<K=720 packed across taps,5 channel tiles>"; break; }
        const int cin = op.in.C + (op.in2.buf >= 0 ? op.in2.C : 0);
        const ConvLaunch c = conv_launch_of(e, op, n);
        if (c.ws1) k = std::string("conv1x1_ws_kernel<") + (op.in2.buf >= 0 ? "dual-source," : "") + "NK=" + std::to_string(cin / 32) + "," + std::to_string(c.ct) + " channel tiles," + std::to_string(c.pt * 16) + " px>";
        else if (c.ps) k = std::string("conv3x3_ws_kernel<") + (op.stride == 2 ? "S=2," : "") + (c.rowt ? "ROWT," : "") + (c.tpw1 ? "TPW=1," : "TPW=2,") + std::to_string(c.ct) + " channel tiles>";
        else if (c.lds) k = "conv3x3_lds_kernel<S=" + std::to_string(op.stride) + ",CT=" + std::to_string(c.ct) + ",PT=" + std::to_string(c.pt) + (c.wres ? ",wres>" : ">");
        else if (c.stream) k = "conv1x1_stream_kernel<CT=" + std::to_string(c.ct) + ",PT=" + std::to_string(c.pt) + ",NK=" + std::to_string((cin + 31) / 32) + ">";
        else k = std::string("conv_igemm_kernel<") + (op.ks == 1 ? (op.in2.buf >= 0 ? "1x1 dual-source" : "1x1") : (c.fastk ? "3x3" : "3x3 generic-K")) +
                 ",CT=" + std::to_string(c.ct) + ",PT=" + std::to_string(c.pt) + ",KSPLIT=" + std::to_string(c.ksplit) + ">";
        break;
    }
    default: k = "?";
    }
    snprintf(out, cap, "%s", k.c_str());
    return ZLY_OK;
}

int32_t zly_launch_info_at(zly_engine* e, int32_t i, int32_t n, zly_launch_info* out)
{
    if (!e || !out || i < 0 || i >= (int32_t)e->ops.size() || n < 1 || n > e->cfg.max_batch) return fail(ZLY_ERR_INVALID_ARGUMENT, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    memset(out, 0, sizeof *out);
    const int nops = (int)e->ops.size();
    out->covered_by = op_covered_by(e, i, n);
    if (out->covered_by != i) return ZLY_OK;
    std::vector<char> in_group((size_t)nops, 0);
    for (int j = 0; j < nops; ++j) in_group[(size_t)j] = op_covered_by(e, j, n) == i ? 1 : 0;
    // channel maps per buffer: written inside the group / read outside the group
    std::map<int, std::vector<char>> written_in, read_out, counted_in, counted_out;
    auto chan = [&](std::map<int, std::vector<char>>& m, int buf) -> std::vector<char>& {
        std::vector<char>& v = m[buf];
        if (v.empty()) v.assign((size_t)e->bufs[(size_t)buf].C + 64, 0);
        return v;
    };
    std::vector<IoView> io;
    for (int j = 0; j < nops; ++j) {
        io.clear();
        if (in_group[(size_t)j]) { op_writes(e, e->ops[(size_t)j], &io); for (const IoView& v : io) for (int c = 0; c < v.C; ++c) chan(written_in, v.buf)[(size_t)(v.co + c)] = 1; }
        else { op_reads(e, e->ops[(size_t)j], &io); for (const IoView& v : io) for (int c = 0; c < v.C; ++c) chan(read_out, v.buf)[(size_t)(v.co + c)] = 1; }
    }
    double ext_in = 0, ext_out = 0, wbytes = 0;
    for (int j = 0; j < nops; ++j) {
        if (!in_group[(size_t)j]) continue;
        const Op& op = e->ops[(size_t)j];
        out->n_ops++;
        out->flops_per_frame += op.flops;
        out->bytes_unfused_per_frame += op.bytes;
        wbytes += op.wbytes;
        if (op.kind == OP_PREPROCESS) ext_in += (double)e->cfg.model_w * e->cfg.model_h * 3;          // the u8 frame (model-sized requests)
        if (op.kind == OP_HEAD && !(e->cfg.flags & ZLY_FLAG_NO_HEAD_TENSOR)) ext_out += (double)op.head.lv[op.level].hw * (4 + e->nc) * 4.0;
        io.clear(); op_reads(e, op, &io);
        for (const IoView& v : io) {
            int cext = 0;
            for (int c = 0; c < v.C; ++c) {
                const size_t ch = (size_t)(v.co + c);
                if (chan(written_in, v.buf)[ch] || chan(counted_in, v.buf)[ch]) continue;
                chan(counted_in, v.buf)[ch] = 1; ++cext;
            }
            ext_in += v.px * cext * (double)e->esz * v.scale;
        }
        io.clear(); op_writes(e, op, &io);
        for (const IoView& v : io) {
            int cext = 0;
            for (int c = 0; c < v.C; ++c) {
                const size_t ch = (size_t)(v.co + c);
                if (!chan(read_out, v.buf)[ch] || chan(counted_out, v.buf)[ch]) continue;
                chan(counted_out, v.buf)[ch] = 1; ++cext;
            }
            ext_out += v.px * cext * (double)e->esz * v.scale;
        }
    }
    out->bytes_fused_per_frame = ext_in + ext_out + wbytes;
    out->weight_bytes = wbytes;
    return ZLY_OK;
}

int32_t zly_profile_ops(zly_engine* e, int32_t n, const void* d_frames, int32_t w, int32_t h, int32_t reps, float* ms_out)
{
    if (!e) return fail(ZLY_ERR_NOT_INITIALIZED, "Engine not running");
    if (!d_frames || !ms_out || reps < 1 || n < 1 || n > e->cfg.max_batch) return fail(ZLY_ERR_INVALID_ARGUMENT, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    ExclusiveGate gl;
    HIP_TRY(hipSetDevice(e->dev), ZLY_ERR_SYSTEM);
    std::vector<int32_t> ws((size_t)n, w), hs((size_t)n, h);
    std::vector<size_t> offs((size_t)n);
    for (int i = 0; i < n; ++i) offs[(size_t)i] = (size_t)i * (size_t)w * (size_t)h * 3u;
    int rc = set_desc(e, n, ws.data(), hs.data(), offs.data(), e->stream);
    if (rc != ZLY_OK) return rc;
    const size_t nops = e->ops.size();
    std::vector<hipEvent_t> ev(nops + 1);
    for (hipEvent_t& x : ev) HIP_TRY(hipEventCreate(&x), ZLY_ERR_SYSTEM);
    std::vector<double> acc(nops, 0.0);
    int rcode = ZLY_OK;
    // ZLY_PROFILE_INNER=k: every op is launched k times back to back between its two events, and the time is
    // divided by k -- the in-sequence cost of a launch (kernel + boundary) without the ~5 us that an event pair
    // per launch adds; used to study the batch-1 path, where most kernels are shorter than that
    int inner = 1;
    if (const char* v = getenv("ZLY_PROFILE_INNER")) inner = atoi(v) > 0 ? atoi(v) : 1;
    for (int r = 0; r < reps && rcode == ZLY_OK; ++r) {
        hipEventRecord(ev[0], e->stream);
        for (size_t i = 0; i < nops; ++i) {
            hipError_t hr = hipSuccess;
            // only the idempotent forward ops are repeated: the Detect tail appends candidates and NMS consumes them
            const int reps_i = (e->ops[i].kind == OP_HEAD || e->ops[i].kind == OP_NMS) ? 1 : inner;
            for (int k = 0; k < reps_i && hr == hipSuccess; ++k) {
                if (e->stem_fused && i == 0) {
                    // shipped path: preprocess is part of the stem kernel; its time is booked on ops[1] (model.0)
                } else if (e->stem1 && i == 1) {
                    Stem1Args st = e->stem1a;
                    st.st.src = (const uint8_t*)d_frames; st.st.desc = e->d_desc;
                    hr = launch_stem_model1(st, n, e->stream);
                } else if (e->stem1 && i == 2) {
                    // model.1 ran inside the stem kernel
                } else if (e->stem_fused && i == 1) {
                    StemArgs st = e->stem;
                    st.src = (const uint8_t*)d_frames; st.desc = e->d_desc;
                    hr = launch_stem_fused(st, n, e->stream);
                } else {
                    hr = run_op(e, e->ops[i], n, (const uint8_t*)d_frames, nullptr, 0, e->stream);
                }
            }
            if (hr != hipSuccess) { rcode = fail(ZLY_ERR_INFERENCE, std::string("profile: ") + hipGetErrorString(hr)); break; }
            hipEventRecord(ev[i + 1], e->stream);
        }
        if (rcode != ZLY_OK) break;
        if (hipStreamSynchronize(e->stream) != hipSuccess) { rcode = fail(ZLY_ERR_INFERENCE, "profile: sync failed"); break; }
        for (size_t i = 0; i < nops; ++i) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            if (op_is_noop(e, e->ops[i], n)) ms = 0.f;       // nothing was launched: what the event pair shows is its own cost
            acc[i] += ms / ((e->ops[i].kind == OP_HEAD || e->ops[i].kind == OP_NMS) ? 1 : inner);
        }
    }
    for (hipEvent_t& x : ev) hipEventDestroy(x);
    if (rcode != ZLY_OK) return rcode;
    for (size_t i = 0; i < nops; ++i) {
        ms_out[i] = (float)(acc[i] / reps);
        const int k = e->ops[i].kind;
        with_stats(e, [&](zly_stats& st) {
            if (k == OP_PREPROCESS) st.total_preprocess_ms += acc[i];
            else if (k == OP_NMS) st.total_postprocess_ms += acc[i];
            else st.total_forward_ms += acc[i];
        });
    }
    e->last_n = n; e->last_stem1 = e->stem1;
    return ZLY_OK;
}

int32_t zly_get_stats(const zly_engine* e, zly_stats* out)
{
    if (!e || !out) return fail(ZLY_ERR_INVALID_ARGUMENT, "bad argument");
    std::lock_guard<std::mutex> lk(e->stats_mu);           // never held across a device call: does not wait on a running batch
    *out = e->stats;
    return ZLY_OK;
}

}  // extern "C"
