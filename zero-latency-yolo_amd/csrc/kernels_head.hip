// kernels_head.hip -- the whole YOLOv8 Detect tail in ONE launch for all three pyramid levels:
//   final 1x1 convs of the box branch (c2 -> 4*16 DFL logits) and class branch (c3 -> nc logits) as
//   two small MFMA GEMMs per 16-anchor tile, then, straight from the fp32 accumulators,
//   DFL softmax-expectation, dist2bbox (xywh) * stride, class sigmoid  -> rows of the [4+nc][N] fp32
//   head tensor (the "output0" tensor of the reference, onnx_engine.cpp:50,767-796), and the
//   reference's decode + confidence threshold (onnx_engine.cpp:779-816) with wavefront-ballot
//   compaction into the per-frame candidate list that the NMS kernel consumes.
// It replaces 6 conv launches + 3 head launches + a memset + the decode launch, and the fp32 logit
// round trip through HBM.  decode here and decode_kernel (kernels_post.hip, used by zly_postprocess)
// apply the same comparisons to the same fp32 values, so both give the same candidates.
#include "zly_internal.h"
#include <algorithm>
#include <math.h>

#pragma clang fp contract(off)

namespace zly {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename T> struct HFrag;
template <> struct HFrag<bf16_t> { typedef bf16x8 type; static constexpr int EPL = 8; static constexpr int KSTEP = 32; };
template <> struct HFrag<float>  { typedef f32x4  type; static constexpr int EPL = 4; static constexpr int KSTEP = 16; };

__device__ __forceinline__ f32x4 hmma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 hmma(f32x4 a, f32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
}

// exp / reciprocal of the DFL softmax and the class sigmoid.  fp32 engine (verification mode): libm expf and IEEE divides.
// bf16 engine: v_exp_f32 / v_rcp_f32 (1 ulp) -- the 36 expf + 24 divides per lane were half of this kernel's time
// (~900 VALU instructions per 16 anchors); the inputs carry bf16 rounding noise four orders of magnitude larger.
// Whatever is computed here is what goes into the head tensor AND into the threshold/arg-max, so detect() stays
// bit-identical to the oracle's post-processing of the engine's own head tensor.
template <typename T> __device__ __forceinline__ float h_exp(float x);
template <> __device__ __forceinline__ float h_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ float h_exp<bf16_t>(float x) { return __builtin_amdgcn_exp2f(x * 1.442695041f); }
template <typename T> __device__ __forceinline__ float h_div(float a, float b);
template <> __device__ __forceinline__ float h_div<float>(float a, float b) { return a / b; }
template <> __device__ __forceinline__ float h_div<bf16_t>(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// A workgroup (HEAD_WAVES = 8 waves) owns HEAD_GROUP = 128 consecutive anchors of one level of one frame, one wave per 16 anchors (MFMA
// columns): lane (p = lane & 15, kq = lane >> 4) ends up holding, for every 16-channel tile c, channels c*16 + kq*4 + {0..3} of anchor p.
// What the ablation showed (tools/head_bench.hip, profiles/r03_head_kernel_ablation.txt; 36.6 us for the batch-64 tail): 11.5 us are the
// 56 832 four-wave... one-tile waves themselves (dispatch + set-up), ~8 us each the weight fragments (23 KiB per wave through L1) and the
// activations, ~10 us the epilogue.  So: the level's two weight matrices go to LDS once per workgroup (23 KiB per 128 anchors instead of
// per 16) and the fragments come from there; the class-branch activations are requested BEFORE the weight staging and its barrier, so
// they are in flight meanwhile; and the epilogue is skipped where it cannot matter:
// Early out (production: no head tensor, no logit dump): when NO anchor of the tile has a class logit above skip_logit =
// logit(conf_thr) - 1e-2, none can reach the confidence threshold (the margin is four orders of magnitude above the error of
// v_exp / v_rcp), and the box loads, the box GEMM, the DFL and the 80 sigmoids per anchor are skipped -- ~4 of 5 tiles on a detector that
// passes ~1.5 % of its anchors.  Tiles that are not skipped run the full arithmetic, so the candidates are exactly the same.
template <typename T, int CTC>
__global__ __launch_bounds__(HEAD_WAVES * 64) void head_fused_kernel(const HeadArgs a)
{
    typedef typename HFrag<T>::type F;
    constexpr int EPL = HFrag<T>::EPL, KSTEP = HFrag<T>::KSTEP, WTILE = 16 * KSTEP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, kq = lane >> 4;
    const int f = blockIdx.y;
    const int bx = blockIdx.x + (a.only_level >= 0 ? a.lv[a.only_level].block0 : 0);
    const int li = bx >= a.lv[2].block0 ? 2 : (bx >= a.lv[1].block0 ? 1 : 0);
    const HeadLevel& L = a.lv[li];
    const int anchor0 = (bx - L.block0) * HEAD_GROUP + wave * 16;
    const bool active = anchor0 < L.hw;                    // wave-uniform; inactive waves still help staging the weights
    const int an = anchor0 + p;
    const bool valid = active && an < L.hw;
    const size_t pix = (size_t)f * L.hw + (valid ? an : 0);

    F zero;
#pragma unroll
    for (int j = 0; j < EPL; ++j) zero[j] = (T)0.0f;
    // the two branch tensors of this level as buffer resources (a.buf32: every level below 2 GiB, set by launch_head_fused)
    const __amdgpu_buffer_rsrc_t rcls = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(L.cls_in), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbox = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(L.box_in), 0, 0x7fffffff, 0x00020000);

    constexpr int KMAX = HEAD_KMAX;                          // k-steps of a branch: bf16 80/128 channels -> 3/4, fp32 -> 5/8 (launch_head_fused refuses more)
    // weights -> LDS: [box: 4 x nkb tiles][class: CTC x nkc tiles], 1 KiB each, already in MFMA lane order
    T* lwb = reinterpret_cast<T*>(smem);
    T* lwc = lwb + (size_t)4 * L.nkb * WTILE;
    const int nb16 = 4 * L.nkb * 64, nc16 = CTC * L.nkc * 64;          // 16-byte units
    const uint4* gb = reinterpret_cast<const uint4*>(L.wb);
    const uint4* gc = reinterpret_cast<const uint4*>(L.wc);
    uint4* sb4 = reinterpret_cast<uint4*>(lwb);
    uint4* sc4 = reinterpret_cast<uint4*>(lwc);
    // every unit of this thread requested before the first LDS store: `for (u ...) lds[u] = g[u]` compiled to load -> s_waitcnt vmcnt(0) -> ds_write per
    // iteration, six global round trips in a row at the head of every workgroup (2 + 4 iterations for the 64 / 80-channel branches).  Buffer loads:
    // units beyond the tile count are out of range (no memory access), the stores stay guarded.
    constexpr int NT = HEAD_WAVES * 64;
    constexpr int NB_T = (4 * KMAX * 64 + NT - 1) / NT, NC_T = (CTC * KMAX * 64 + NT - 1) / NT;
    const __amdgpu_buffer_rsrc_t rwb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(L.wb), 0, (unsigned)nb16 * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(L.wc), 0, (unsigned)nc16 * 16u, 0x00020000);
    (void)gb; (void)gc;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    u32x4 tb[NB_T], tc[NC_T];
#pragma unroll
    for (int i = 0; i < NB_T; ++i) if (i * NT < nb16) tb[i] = __builtin_amdgcn_raw_buffer_load_b128(rwb, (int)((threadIdx.x + i * NT) * 16), 0, 0);      // scalar test: whole rounds beyond the branch's tiles are skipped
#pragma unroll
    for (int i = 0; i < NC_T; ++i) if (i * NT < nc16) tc[i] = __builtin_amdgcn_raw_buffer_load_b128(rwc, (int)((threadIdx.x + i * NT) * 16), 0, 0);
    // (the weight loads above and the class fragments below are in flight together; the LDS stores of the weights follow the fragment requests)
    // class-branch fragments of the tile, every k-step, requested up front
    F xc[KMAX];
    {
        const T* pc = static_cast<const T*>(L.cls_in) + pix * L.cls_cs;
        // No branch around a load: `if (valid && ...) x = *p` compiled to exec-masked blocks with `s_waitcnt vmcnt(0)` behind every second load -- the k-steps
        // of a tile were fetched in a chain of global round trips at the head of every workgroup.  With 32-bit offsets available (a.buf32, every realistic
        // batch) the lane mask goes into the offset of a buffer load (out of range -> zeros) and only the k-step test stays, as a scalar branch.
        if (a.buf32) {
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                const int ci = s * KSTEP + kq * EPL;
                xc[s] = zero;
                if (s < L.nkc) {
                    const unsigned off = (valid && ci < L.cls_cin) ? (unsigned)((pix * L.cls_cs + ci) * sizeof(T)) : 0x80000000u;
                    xc[s] = __builtin_bit_cast(F, __builtin_amdgcn_raw_buffer_load_b128(rcls, (int)off, 0, 0));
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                const int ci = s * KSTEP + kq * EPL;
                xc[s] = zero;
                if (s < L.nkc && valid && ci < L.cls_cin) xc[s] = *reinterpret_cast<const F*>(pc + ci);
            }
        }
    }
    // class biases of this lane's rows: requested with everything else (loaded where they are added they were one more round trip between the class GEMM and the early-out)
    f32x4 bcls[CTC];
#pragma unroll
    for (int c = 0; c < CTC; ++c) bcls[c] = *reinterpret_cast<const f32x4*>(L.bc + c * 16 + kq * 4);
#pragma unroll
    for (int i = 0; i < NB_T; ++i) { const int u = threadIdx.x + i * NT; if (i * NT < nb16 && u < nb16) *reinterpret_cast<u32x4*>(sb4 + u) = tb[i]; }
#pragma unroll
    for (int i = 0; i < NC_T; ++i) { const int u = threadIdx.x + i * NT; if (i * NT < nc16 && u < nc16) *reinterpret_cast<u32x4*>(sc4 + u) = tc[i]; }
    __syncthreads();
    if (!active) return;

    // ---- class branch: [nc x cin] . [cin x 16] ---------------------------------------------------
    f32x4 accc[CTC];
#pragma unroll
    for (int c = 0; c < CTC; ++c) accc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const T* w = lwc + lane * EPL;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            if (s < L.nkc) {
#pragma unroll
                for (int c = 0; c < CTC; ++c) {
                    const F wf = *reinterpret_cast<const F*>(w + ((size_t)c * L.nkc + s) * WTILE);
                    accc[c] = hmma(wf, xc[s], accc[c]);
                }
            }
        }
    }
    if (a.head == nullptr && L.logits == nullptr) {
        float zmax = -3.0e38f;
#pragma unroll
        for (int c = 0; c < CTC; ++c) {
            const int ch = c * 16 + kq * 4;
            const f32x4 bias = bcls[c];
#pragma unroll
            for (int r = 0; r < 4; ++r) if (ch + r < a.nc) zmax = fmaxf(zmax, accc[c][r] + bias[r]);
        }
        zmax = fmaxf(zmax, __shfl_xor(zmax, 16));
        zmax = fmaxf(zmax, __shfl_xor(zmax, 32));
        if (__ballot(valid && zmax >= a.skip_logit) == 0ull) return;
    }
    // ---- box branch: [64 x cin] . [cin x 16] ----------------------------------------------------
    f32x4 accb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) accb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        F xb[KMAX];
        const T* pb = static_cast<const T*>(L.box_in) + pix * L.box_cs;
        if (a.buf32) {
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                const int ci = s * KSTEP + kq * EPL;
                xb[s] = zero;
                if (s < L.nkb) {
                    const unsigned off = (valid && ci < L.box_cin) ? (unsigned)((pix * L.box_cs + ci) * sizeof(T)) : 0x80000000u;
                    xb[s] = __builtin_bit_cast(F, __builtin_amdgcn_raw_buffer_load_b128(rbox, (int)off, 0, 0));
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                const int ci = s * KSTEP + kq * EPL;
                xb[s] = zero;
                if (s < L.nkb && valid && ci < L.box_cin) xb[s] = *reinterpret_cast<const F*>(pb + ci);
            }
        }
        const T* w = lwb + lane * EPL;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            if (s < L.nkb) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const F wf = *reinterpret_cast<const F*>(w + ((size_t)c * L.nkb + s) * WTILE);
                    accb[c] = hmma(wf, xb[s], accb[c]);
                }
            }
        }
    }

    // ---- DFL: side c (l,t,r,b) = channel tile c; its 16 bins live in the 4 lanes {p, p+16, p+32, p+48} ----
    float dist[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(L.bb + c * 16 + kq * 4);
        const f32x4 v = accb[c] + bias;
        accb[c] = v;
        float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float se = 0.f, sw = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = h_exp<T>(v[r] - m);
            se += e;
            sw += e * (float)(kq * 4 + r);
        }
        se += __shfl_xor(se, 16); sw += __shfl_xor(sw, 16);
        se += __shfl_xor(se, 32); sw += __shfl_xor(sw, 32);
        dist[c] = h_div<T>(sw, se);
    }
    const float ax = (float)(an % L.W) + 0.5f, ay = (float)(an / L.W) + 0.5f;
    const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
    const float sp = (float)L.stride_px;
    const float cx = (x1 + x2) / 2.0f * sp, cy = (y1 + y2) / 2.0f * sp, bw = (x2 - x1) * sp, bh = (y2 - y1) * sp;

    // ---- class scores + running arg-max in the reference's order (strict >, lowest class wins ties) ----
    float best = 0.0f;
    int cls = -1;
    f32x4 score[CTC];
#pragma unroll
    for (int c = 0; c < CTC; ++c) {
        const int ch = c * 16 + kq * 4;
        const f32x4 bias = bcls[c];
        const f32x4 z = accc[c] + bias;
        accc[c] = z;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sc = h_div<T>(1.0f, 1.0f + h_exp<T>(-z[r]));
            score[c][r] = sc;
            if (ch + r < a.nc && sc > best) { best = sc; cls = ch + r; }
        }
    }
#pragma unroll
    for (int d = 16; d <= 32; d <<= 1) {
        const float ob = __shfl_xor(best, d);
        const int oc = __shfl_xor(cls, d);
        if (ob > best || (ob == best && oc >= 0 && (cls < 0 || oc < cls))) { best = ob; cls = oc; }
    }

    // ---- outputs ------------------------------------------------------------------------------------
    if (a.head && valid) {
        float* h = a.head + (size_t)f * (4 + a.nc) * a.N_total + L.anchor_off + an;
        const float bv = kq == 0 ? cx : (kq == 1 ? cy : (kq == 2 ? bw : bh));
        h[(size_t)kq * a.N_total] = bv;
#pragma unroll
        for (int c = 0; c < CTC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = c * 16 + kq * 4 + r;
                if (ch < a.nc) h[(size_t)(4 + ch) * a.N_total] = score[c][r];
            }
    }
    if (L.logits && valid) {
        float* lg = L.logits + pix * L.logits_cs;
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(lg + c * 16 + kq * 4) = accb[c];
#pragma unroll
        for (int c = 0; c < CTC; ++c)
            if (c * 16 + kq * 4 < L.logits_cs - 64) *reinterpret_cast<f32x4*>(lg + 64 + c * 16 + kq * 4) = accc[c];
    }
    if (a.cand) {
        // decode + threshold (onnx_engine.cpp:799-816); lanes 0..15 stand for the wave's 16 anchors
        const bool pass = valid && kq == 0 && best >= a.conf_thr && cls >= 0;
        const unsigned long long mask = __ballot(pass);
        if (mask != 0ull) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&a.cand_count[f], __popcll(mask));
            base = __shfl(base, 0);
            if (pass) {
                const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
                if (slot < a.N_total) {
                    const FrameDesc d = a.desc[f];
                    Cand c;
                    c.x = cx / (float)d.w; c.y = cy / (float)d.h; c.w = bw / (float)d.w; c.h = bh / (float)d.h;
                    c.conf = best; c.cls = cls; c.anchor = L.anchor_off + an; c.pad_ = 0;
                    a.cand[(size_t)f * a.N_total + slot] = c;
                }
            }
        }
    }
}

typedef void (*head_fn)(const HeadArgs);
template <typename T> static head_fn pick_head(int ctc) {
    switch (ctc) {
        case 1: return head_fused_kernel<T, 1>;
        case 2: return head_fused_kernel<T, 2>;
        case 3: return head_fused_kernel<T, 3>;
        case 4: return head_fused_kernel<T, 4>;
        case 5: return head_fused_kernel<T, 5>;
    }
    return nullptr;
}

hipError_t launch_head_fused(int dtype, const HeadArgs& a0, int n, hipStream_t s)
{
    HeadArgs a = a0;
    // the class logit below which no score can reach conf_thr (see the kernel): logit(thr) minus a margin; thr <= 0 keeps everything,
    // thr >= 1 needs sigmoid(z) to round to 1.0f, i.e. z > 17
    const double thr = (double)a.conf_thr;
    a.skip_logit = thr <= 0.0 ? -3.0e38f : (thr >= 1.0 ? 15.0f : (float)(log(thr / (1.0 - thr)) - 1e-2));
    const int ctc = (a.nc + 15) / 16;
    head_fn fn = dtype == ZLY_DTYPE_BF16 ? pick_head<bf16_t>(ctc) : pick_head<float>(ctc);
    if (!fn) return hipErrorInvalidValue;                  // nc > 80 is not supported by this kernel
    if (a.only_level > 2) return hipErrorInvalidValue;
    size_t lds = 0;
    for (int l = 0; l < 3; ++l) { const size_t b = (size_t)(4 * a.lv[l].nkb + ctc * a.lv[l].nkc) * 1024; lds = b > lds ? b : lds; }
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    for (int l = 0; l < 3; ++l)
        if (a.lv[l].nkb > HEAD_KMAX || a.lv[l].nkc > HEAD_KMAX) return hipErrorInvalidValue;      // the kernel holds a branch's fragments in HEAD_KMAX k-steps of registers: a wider branch would lose its tail k-steps silently
    a.buf32 = 1;
    for (int l = 0; l < 3; ++l) {
        const size_t el = dtype == ZLY_DTYPE_BF16 ? 2 : 4;
        if ((size_t)n * a.lv[l].hw * (size_t)std::max(a.lv[l].box_cs, a.lv[l].cls_cs) * el >= ((size_t)1 << 31)) a.buf32 = 0;      // 32-bit offsets would not reach: pointer loads
    }
    const int blocks = a.only_level >= 0 ? (a.lv[a.only_level].hw + HEAD_GROUP - 1) / HEAD_GROUP : a.total_blocks;
    hipLaunchKernelGGL(fn, dim3(blocks, n), dim3(HEAD_WAVES * 64), lds, s, a);
    return hipGetLastError();
}

}  // namespace zly
