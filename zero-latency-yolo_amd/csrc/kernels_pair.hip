// kernels_pair.hip -- a C2f bottleneck (3x3 conv -> 3x3 conv [+ shortcut]) as ONE kernel, bf16, C = 16 or 32 channels.
//
// Replaces two conv nodes (+ the Add) that the reference executes inside Ort::Session::Run
// (reference src/inference/onnx_engine.cpp:578-585): ultralytics Bottleneck = cv2(cv1(x)) (+ x), both 3x3, c -> c.
//
// These are the high-resolution layers of the net (104x104 x 16 ch, 52x52 x 32 ch at 416x416): a few MFMAs per pixel,
// so run one at a time each launch is bound by writing its output to HBM and reading it back (and by one more
// launch).  Here a workgroup owns a TH x TW output tile of one frame and keeps the intermediate map in LDS:
//   stage   x patch  (TH+4) x (TW+4) x C   global -> registers (prefetched one tile ahead) -> LDS, zero outside the frame
//   conv A  over the (TH+2) x (TW+2) halo region, bias + SiLU, ZERO outside the frame (conv B pads the intermediate
//           map with zeros, not with conv A of the padded input), rounded to bf16 -> LDS          [same value the
//           unfused path writes to HBM]
//   conv B  over TH x TW from the LDS intermediate, bias + SiLU (+ x from the LDS patch) -> HBM, 16-byte stores
// Both convs' weights (9 KB / 36 KB) are copied to LDS once per workgroup; workgroups are persistent over tiles.
// Pixels of a region are linearised (q -> (q / RW, q % RW)) and cut into 16-pixel MFMA tiles, so a tile shape need
// not be a multiple of 16 wide: 13 x 26, 26 x 26 ... are picked per layer on the host (pair_plan) to divide the map.
// One workgroup per CU (up to 16 waves): phases are separated by workgroup barriers, and inside a phase waves sit at
// different points of their (MFMA loop, SiLU epilogue) sequence, which overlaps the matrix and vector pipes.
// C = 16 uses v_mfma_f32_16x16x16_bf16 (one tap = one k-step, 8-byte fragments, pixel pitch 48 B); C = 32 uses
// v_mfma_f32_16x16x32_bf16 (16-byte fragments, pitch 96 B).  Pitches are conflict-free for the fragment reads
// (tools/lds_pitch.py).
#include "zly_internal.h"
#include "conv_device.h"
#include <stdlib.h>
#include <stdio.h>

namespace zly {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int C> struct PairGeom;
template <> struct PairGeom<16> {
    static constexpr int CT = 1, KS = 1, FRAGB = 8, PITCH = 48, WTILE = 512;
    static constexpr bool WRES = true;
    typedef s16x4 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
};
template <> struct PairGeom<32> {
    static constexpr int CT = 2, KS = 1, FRAGB = 16, PITCH = 96, WTILE = 1024;
    static constexpr bool WRES = true;
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
// C = 64 (the 26x26 stage at 416x416): two k-steps per tap, pixel pitch 160 B (conflict-free for 128-byte pixels), and ONE
// 72 KB weight buffer that holds conv A's weights, then conv B's (both do not fit next to the patches): the kernel is
// about launches here -- two ~9 us launches become one.  Measured at batch 64: 18.5 us fused vs 2 x 9 us (one tile per CU,
// so the patch load and both 72 KB weight loads are exposed back to back); batch 1: +0.5 %.  Off by default
// (ZLY_PAIR_WIDTHS=112 enables it); prefetching conv B's weights during conv A is the obvious next step.
template <> struct PairGeom<64> {
    static constexpr int CT = 4, KS = 2, FRAGB = 16, PITCH = 160, WTILE = 1024;
    static constexpr bool WRES = false;
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ int div_small(int q, float inv) { return (int)(((float)q + 0.5f) * inv); }   // exact for q < 2^20, divisor < 2^10

// 9 taps of one 16-pixel tile: src = LDS image with row pitch `rowb` bytes, `off` = this lane's pixel/k-group offset
template <int C>
__device__ __forceinline__ void pair_taps(const unsigned char* __restrict__ src, const unsigned char* __restrict__ lw, int off, int rowb, int lane,
                                          f32x4 (&acc)[PairGeom<C>::CT])
{
    typedef PairGeom<C> G;
    typedef typename G::frag F;
#pragma unroll
    for (int c = 0; c < G::CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* wl = lw + lane * G::FRAGB;
    // C = 16 (8-byte fragments): the kx offsets as opaque registers, so that the three reads of a row are never fused into ds_read2_b64 (conv_device.h)
    const int kxo[3] = {0, G::FRAGB == 8 ? opaque_offset(G::PITCH) : G::PITCH, G::FRAGB == 8 ? opaque_offset(2 * G::PITCH) : 2 * G::PITCH};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const unsigned char* row = src + off + ky * rowb;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const F x = *reinterpret_cast<const F*>(row + (G::FRAGB == 8 ? kxo[kx] : kx * G::PITCH) + ks * 64);
#pragma unroll
                for (int c = 0; c < G::CT; ++c) {
                    const F w = *reinterpret_cast<const F*>(wl + ((c * 9 + ky * 3 + kx) * G::KS + ks) * G::WTILE);
                    acc[c] = G::mma(w, x, acc[c]);
                }
            }
        }
    }
}

template <int C, int NW, int NLD>
__global__ __launch_bounds__(NW * 64) void bottleneck_pair_kernel(const PairArgs a)
{
    typedef PairGeom<C> G;
    constexpr int NT = NW * 64;
    constexpr int UPP = C / 8;                        // 16-byte units per pixel
    constexpr int WBYTES = 9 * G::KS * G::CT * G::WTILE;   // one conv's weights
    constexpr int WBUFS = G::WRES ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lwA = smem;
    unsigned char* lwB = smem + (WBUFS - 1) * WBYTES;
    unsigned char* lin = smem + WBUFS * WBYTES;
    const int PW = a.TW + 4, PH = a.TH + 4, MW = a.TW + 2, MH = a.TH + 2;
    unsigned char* lmid = lin + (PH * PW * G::PITCH + 15) / 16 * 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kq = lane >> 4;
    const bf16_t* __restrict__ in = static_cast<const bf16_t*>(a.in) + a.in_co;
    bf16_t* __restrict__ out = static_cast<bf16_t*>(a.out) + a.out_co;

    auto stage_weights = [&](unsigned char* dst, const void* src) {
        for (int u = tid; u < WBYTES / 16; u += NT)
            *reinterpret_cast<u32x4_t*>(dst + (size_t)u * 16) = *reinterpret_cast<const u32x4_t*>(static_cast<const unsigned char*>(src) + (size_t)u * 16);
    };
    if (G::WRES) {                                     // weights of both convs -> LDS, once
        stage_weights(lwA, a.wA);
        stage_weights(lwB, a.wB);
    }
    // per-lane bias: C = 16 -> channels kq*4..+3; C >= 32 (pair-permuted rows) -> tiles 2j, 2j+1: j*32 + kq*8 .. +3 / +4..+7
    f32x4 biasA[G::CT], biasB[G::CT];
#pragma unroll
    for (int c = 0; c < G::CT; ++c) {
        const int ch = C == 16 ? kq * 4 : (c >> 1) * 32 + kq * 8 + (c & 1) * 4;
        biasA[c] = *reinterpret_cast<const f32x4*>(a.bA + ch);
        biasB[c] = *reinterpret_cast<const f32x4*>(a.bB + ch);
    }

    const int NPU = PH * PW * UPP;                    // 16-byte units of the x patch (host guarantees NPU <= NT * NLD)
    const float invPW = 1.0f / (float)PW, invMW = 1.0f / (float)MW, invTW = 1.0f / (float)a.TW;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int NPA = MH * MW, NPB = a.TH * a.TW;
    const int ntA = (NPA + 15) >> 4, ntB = (NPB + 15) >> 4;

    auto tile_origin = [&](int tl, int& b, int& y0, int& x0) {
        b = tl / tiles_per_img;
        const int r = tl - b * tiles_per_img;
        const int ty = r / a.tiles_x;
        y0 = ty * a.TH; x0 = (r - ty * a.tiles_x) * a.TW;
    };
    u32x4_t pre[NLD];
    auto stage_load = [&](int tl) {
        int b, y0, x0;
        tile_origin(tl, b, y0, x0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int u = tid + i * NT;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (u < NPU) {
                const int px = u / UPP, part = u - px * UPP;
                const int py = div_small(px, invPW), pxx = px - py * PW;
                const int gy = y0 - 2 + py, gx = x0 - 2 + pxx;
                if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                    v = *reinterpret_cast<const u32x4_t*>(in + ((size_t)(b * a.H + gy) * a.W + gx) * a.in_cs + part * 8);
            }
            pre[i] = v;
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int u = tid + i * NT;
            if (u < NPU) {
                const int px = u / UPP, part = u - px * UPP;
                *reinterpret_cast<u32x4_t*>(lin + px * G::PITCH + part * 16) = pre[i];
            }
        }
    };

    int tl = blockIdx.x;
    if (tl >= a.total_tiles) return;
    stage_load(tl);
    while (true) {
        int b, y0, x0;
        tile_origin(tl, b, y0, x0);
        stage_store();
        if (!G::WRES) stage_weights(lwA, a.wA);        // conv A's weights into the shared weight buffer
        __syncthreads();                               // x patch (and the weights) visible; previous tile's readers done
        const int tnext = tl + gridDim.x;
        if (tnext < a.total_tiles) stage_load(tnext);

        // ---- conv A: x patch -> intermediate map in LDS -------------------------------------------------------
        for (int t = wave; t < ntA; t += NW) {
            // C = 64: keep the 72 weight fragments in LDS -- left alone the compiler hoists the loop-invariant reads into
            // 288 registers (fine for the 18 of C = 32, 295 spills here)
            if (C == 64) asm volatile("" ::: "memory");
            const int q = t * 16 + p;
            const int qc = min(q, NPA - 1);
            const int my = div_small(qc, invMW), mx = qc - my * MW;
            f32x4 acc[G::CT];
            pair_taps<C>(lin, lwA, (my * PW + mx) * G::PITCH + kq * G::FRAGB, PW * G::PITCH, lane, acc);
            const int gy = y0 - 1 + my, gx = x0 - 1 + mx;
            const bool inimg = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            f32x4 o[G::CT];
#pragma unroll
            for (int c = 0; c < G::CT; ++c) {
                f32x4 v = acc[c] + biasA[c];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = inimg ? silu<bf16_t>(v[r]) : 0.0f;
                o[c] = v;
            }
            if (q < NPA) {
                unsigned char* dst = lmid + (my * MW + mx) * G::PITCH;
                if (C == 16) store4(reinterpret_cast<bf16_t*>(dst) + kq * 4, o[0]);
                else {
#pragma unroll
                    for (int g2 = 0; g2 < G::CT / 2; ++g2) store8(reinterpret_cast<bf16_t*>(dst) + g2 * 32 + kq * 8, o[2 * g2], o[(2 * g2 + 1) % G::CT]);
                }
            }
        }
        __syncthreads();                               // intermediate map complete (and conv A's weights no longer needed)
        if (!G::WRES) {
            stage_weights(lwB, a.wB);                  // conv B's weights over conv A's
            __syncthreads();
        }

        // ---- conv B: intermediate -> output (+ shortcut from the x patch) --------------------------------------
        for (int t = wave; t < ntB; t += NW) {
            if (C == 64) asm volatile("" ::: "memory");
            const int q = t * 16 + p;
            const int qc = min(q, NPB - 1);
            const int oy = div_small(qc, invTW), ox = qc - oy * a.TW;
            f32x4 acc[G::CT];
            pair_taps<C>(lmid, lwB, (oy * MW + ox) * G::PITCH + kq * G::FRAGB, MW * G::PITCH, lane, acc);
            f32x4 o[G::CT];
#pragma unroll
            for (int c = 0; c < G::CT; ++c) {
                f32x4 v = acc[c] + biasB[c];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu<bf16_t>(v[r]);
                o[c] = v;
            }
            const int gy = y0 + oy, gx = x0 + ox;
            if (q < NPB && gy < a.H && gx < a.W) {
                const unsigned char* xs = lin + ((oy + 2) * PW + ox + 2) * G::PITCH;
                bf16_t* dst = out + ((size_t)(b * a.H + gy) * a.W + gx) * a.out_cs;
                if (C == 16) {
                    if (a.res) o[0] += load4(reinterpret_cast<const bf16_t*>(xs) + kq * 4);
                    store4(dst + kq * 4, o[0]);
                } else {
#pragma unroll
                    for (int g2 = 0; g2 < G::CT / 2; ++g2) {
                        f32x4 lo = o[2 * g2], hi = o[(2 * g2 + 1) % G::CT];
                        if (a.res) {
                            f32x4 ra, rb;
                            load8(reinterpret_cast<const bf16_t*>(xs) + g2 * 32 + kq * 8, ra, rb);
                            lo += ra; hi += rb;
                        }
                        store8(dst + g2 * 32 + kq * 8, lo, hi);
                    }
                }
            }
        }
        if (tnext >= a.total_tiles) break;
        tl = tnext;
        __syncthreads();                               // all reads of the x patch / intermediate done before they are overwritten
    }
}

// ------------------------------------------------------------------------------------------------
// C2f block around one bottleneck as ONE kernel (bf16, C = 16 / 32): ultralytics C2f = cv2(cat(chunk(cv1(x)), m_i(...))).
//
//   MODE 3  cv1 (1x1) -> bottleneck -> cv2 (1x1)      a whole C2f with one bottleneck (model.2, model.15): 3 launches -> 1,
//                                                     the concat buffer never leaves LDS
//   MODE 1  cv1 (1x1) -> bottleneck                   front half of a C2f with two bottlenecks (model.4): y0 | y1 | y2 go to
//                                                     the concat buffer in HBM for MODE 2
//   MODE 2  bottleneck -> cv2 (1x1)                   back half: y0 | y1 come from the concat buffer, y2 is the staged patch,
//                                                     y3 stays in LDS
// cv1 is pointwise, so it is simply evaluated on every pixel of the bottleneck's (TH+4) x (TW+4) input patch straight from
// global memory (fragments = 16-byte NHWC loads, optionally dual-source: the fused Upsample+Concat of the neck), its second
// half y1 lands in the patch buffer the 3x3 convs read, its first half y0 (tile interior only) in an LDS map for cv2.  cv2
// takes its k-steps from the LDS maps in concat order.  Every intermediate is rounded to bf16 exactly where the unfused
// path rounds it (bias + SiLU in fp32, then bf16), so the results match the one-kernel-per-conv path up to fp32 summation
// order.  Halo pixels of cv1 are recomputed by neighbouring tiles (1.3-1.7x the cv1 work; these layers are bound by bytes
// and SiLU issue, not by MFMAs).  With a.dump (debug taps) the intermediates are also written to the concat buffer.
// ------------------------------------------------------------------------------------------------
#ifdef ZLY_C2F_DIAG
__device__ unsigned long long* g_c2f_diag = nullptr;             // diagnostic build only (tools/c2f_bench.hip): per-wave cycle sums of the phases
#define C2FSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); dsum[k] += t_ - dT0; dT0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define C2FSTAMP(k) do { } while (0)
#endif
static constexpr int C2F_BIAS_BYTES = 1024;
template <int C, int MODE, int NW, int NLD, int NK1>
__global__ __launch_bounds__(NW * 64) void c2f_kernel(const C2fArgs a)
{
    typedef PairGeom<C> G;
    typedef typename G::frag F;
    constexpr bool FRONT = (MODE & 1) != 0, BACK = (MODE & 2) != 0;
    constexpr int NT = NW * 64;
    constexpr int UPP = C / 8;
    constexpr int WBYTES = 9 * G::CT * G::WTILE;          // one 3x3 conv's weights
    constexpr int T1 = 2 * C / 16;                        // cv1 output tiles (y0 | y1)
    constexpr int T2MAX = 4;                              // cv2 output tiles (Cout2 = 32 or 64)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef ZLY_C2F_DIAG
    unsigned long long dsum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dT0 = __builtin_amdgcn_s_memtime();
    const unsigned long long dstart = dT0;
#endif
    const int PW = a.TW + 4, PH = a.TH + 4, MW = a.TW + 2, MH = a.TH + 2;
    const int w1_bytes = FRONT ? T1 * a.nk1 * 1024 : 0;
    const int T2 = a.Cout2 >> 4;
    const int w2_bytes = BACK ? T2 * a.nk2 * G::WTILE : 0;
    // biases live in LDS (1 KiB: bA | bB | b1 | b2) and are read where an epilogue needs them: as registers they were 64 VGPRs held through the
    // whole kernel, which is what kept the 32-channel kernel from running 16 waves (128 VGPRs) without spilling
    float* lbias = reinterpret_cast<float*>(smem);
    unsigned char* lwA = smem + C2F_BIAS_BYTES;
    unsigned char* lwB = lwA + WBYTES;
    unsigned char* lw1 = lwB + WBYTES;
    unsigned char* lw2 = lw1 + w1_bytes;
    unsigned char* lin = lw2 + w2_bytes;
    unsigned char* lmid = lin + (PH * PW * G::PITCH + 15) / 16 * 16;
    unsigned char* ly2 = lmid + (MH * MW * G::PITCH + 15) / 16 * 16;                       // BACK: the bottleneck's output map
    unsigned char* ly0 = ly2 + (BACK ? (a.TH * a.TW * G::PITCH + 15) / 16 * 16 : 0);      // MODE 3: first half of cv1's output

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kq = lane >> 4;
    bf16_t* __restrict__ cat = static_cast<bf16_t*>(a.cat);

    auto stage = [&](unsigned char* dst, const void* src, int bytes) {
        for (int u = tid; u < bytes / 16; u += NT)
            *reinterpret_cast<u32x4_t*>(dst + (size_t)u * 16) = *reinterpret_cast<const u32x4_t*>(static_cast<const unsigned char*>(src) + (size_t)u * 16);
    };
    stage(lwA, a.wA, WBYTES);
    stage(lwB, a.wB, WBYTES);
    if (FRONT) stage(lw1, a.w1, w1_bytes);
    if (BACK) stage(lw2, a.w2, w2_bytes);

    constexpr bool BIAS_LDS = C == 32 && NW == 16;     // 8-wave builds keep them in registers (an LDS read per epilogue cost them 2-4 us per launch)
    if (BIAS_LDS) {
        for (int u = tid; u < C; u += NT) { lbias[u] = a.bA[u]; lbias[C + u] = a.bB[u]; }
        if (FRONT) for (int u = tid; u < 2 * C; u += NT) lbias[2 * C + u] = a.b1[u];
        if (BACK) for (int u = tid; u < a.Cout2; u += NT) lbias[4 * C + u] = a.b2[u];
    }
    // per-lane bias of MFMA tile c: C = 16 -> channels kq*4..+3 (cv1: c*16 + kq*4); C = 32 (pair-permuted rows) -> tiles 2j, 2j+1: j*32 + kq*8 .. +3 / +4..+7
    constexpr int NBR = BIAS_LDS ? 1 : G::CT, NB1 = BIAS_LDS ? 1 : T1, NB2 = BIAS_LDS ? 1 : T2MAX;
    f32x4 rbA[NBR], rbB[NBR], rb1[NB1], rb2[NB2];
    if (!BIAS_LDS) {
#pragma unroll
        for (int c = 0; c < G::CT; ++c) {
            const int ch = C == 16 ? kq * 4 : (c >> 1) * 32 + kq * 8 + (c & 1) * 4;
            rbA[c % NBR] = *reinterpret_cast<const f32x4*>(a.bA + ch);
            rbB[c % NBR] = *reinterpret_cast<const f32x4*>(a.bB + ch);
        }
#pragma unroll
        for (int c = 0; c < T1; ++c) {
            const int ch = C == 16 ? c * 16 + kq * 4 : (c >> 1) * 32 + kq * 8 + (c & 1) * 4;
            rb1[c % NB1] = FRONT ? *reinterpret_cast<const f32x4*>(a.b1 + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < T2MAX; ++c)
            rb2[c % NB2] = (BACK && c < T2) ? *reinterpret_cast<const f32x4*>(a.b2 + (c >> 1) * 32 + kq * 8 + (c & 1) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto bias_of = [&](int which, int c) -> f32x4 {       // which: 0 = conv A, 1 = conv B, 2 = cv1, 3 = cv2
        if (BIAS_LDS) {
            const int base = which == 0 ? 0 : which == 1 ? C : which == 2 ? 2 * C : 4 * C;
            return *reinterpret_cast<const f32x4*>(lbias + base + (c >> 1) * 32 + kq * 8 + (c & 1) * 4);
        }
        return which == 0 ? rbA[c % NBR] : which == 1 ? rbB[c % NBR] : which == 2 ? rb1[c % NB1] : rb2[c % NB2];
    };
    const float invPW = 1.0f / (float)PW, invMW = 1.0f / (float)MW, invTW = 1.0f / (float)a.TW;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int NP0 = PH * PW, NPA = MH * MW, NPB = a.TH * a.TW;
    const int nt0 = (NP0 + 15) >> 4, ntA = (NPA + 15) >> 4, ntB = (NPB + 15) >> 4;
    const int NPU = NP0 * UPP;

    auto tile_origin = [&](int tl, int& b, int& y0, int& x0) {
        b = tl / tiles_per_img;
        const int r = tl - b * tiles_per_img;
        const int ty = r / a.tiles_x;
        y0 = ty * a.TH; x0 = (r - ty * a.tiles_x) * a.TW;
    };
    // one map pixel (C channels) <- the CT accumulator tiles of a lane (same layout conv A writes)
    auto put_map = [&](unsigned char* dst, const f32x4* o) {
        if (C == 16) store4(reinterpret_cast<bf16_t*>(dst) + kq * 4, o[0]);
        else store8(reinterpret_cast<bf16_t*>(dst) + kq * 8, o[0], o[1]);
    };
    // stores into the concat buffer go through a buffer resource: one 32-bit element offset per store instead of a 64-bit address pair (the
    // 16-wave build has 128 VGPRs)
    const __amdgpu_buffer_rsrc_t rcat = __builtin_amdgcn_make_buffer_rsrc(a.cat, 0, (unsigned)((size_t)a.n * a.H * a.W * a.cat_cs * 2), 0x00020000);
    auto put_global = [&](int elem, const f32x4* o) {
        if (C == 16) {
            const bf16x4 v = to_bf16x4(o[0]);
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), rcat, (elem + kq * 4) * 2, 0, 0);
        } else {
            const bf16x8 v = to_bf16x8(o[0], o[1]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rcat, (elem + kq * 8) * 2, 0, 0);
        }
    };

    // MODE 2: the bottleneck's input patch comes from the concat buffer (prefetched one tile ahead, as bottleneck_pair_kernel does)
    u32x4_t pre[NLD];
    auto stage_load = [&](int tl) {
        int b, y0, x0;
        tile_origin(tl, b, y0, x0);
        const bf16_t* in = cat + a.pair_in_co;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int u = tid + i * NT;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (u < NPU) {
                const int px = u / UPP, part = u - px * UPP;
                const int py = div_small(px, invPW), pxx = px - py * PW;
                const int gy = y0 - 2 + py, gx = x0 - 2 + pxx;
                if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                    v = *reinterpret_cast<const u32x4_t*>(in + ((size_t)(b * a.H + gy) * a.W + gx) * a.cat_cs + part * 8);
            }
            pre[i] = v;
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int u = tid + i * NT;
            if (u < NPU) {
                const int px = u / UPP, part = u - px * UPP;
                *reinterpret_cast<u32x4_t*>(lin + px * G::PITCH + part * 16) = pre[i];
            }
        }
    };

    int tl = blockIdx.x;
    if (tl >= a.total_tiles) return;
    if (!FRONT) stage_load(tl);
    C2FSTAMP(0);                                       // prologue: weight staging issued, biases, first patch loads issued
    while (true) {
        int b, y0, x0;
        tile_origin(tl, b, y0, x0);
        const int tnext = tl + gridDim.x;
        if (!FRONT) {
            stage_store();
            __syncthreads();                           // patch (and, first time, the weights) visible; previous tile's readers done
            C2FSTAMP(1);                               // patch store + barrier
            if (tnext < a.total_tiles) stage_load(tnext);
        } else {
            // ---- cv1 on every pixel of the patch: x (global) -> y0 | y1 ----------------------------------------------
            const bf16_t* __restrict__ xa = static_cast<const bf16_t*>(a.x) + a.x_co;
            const bf16_t* __restrict__ xb = static_cast<const bf16_t*>(a.x2) + a.x2_co;
            const bool dual = a.x2 != nullptr;
            const unsigned char* w1l = lw1 + lane * 16;
            // the NK1 input fragments of a 16-pixel tile are loaded together, one tile ahead of the MFMAs that consume them: with one
            // load per k-step every k-step exposed an L2 round trip (model.15.cv1: 6 per tile, most of the kernel's time)
            auto load_x = [&](int t, bf16x8 (&xf)[NK1]) {
                const int qc = min(t * 16 + p, NP0 - 1);
                const int py = div_small(qc, invPW), px = qc - py * PW;
                const int gy = y0 - 2 + py, gx = x0 - 2 + px;
                const bool inimg = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const size_t pa = dual ? ((size_t)(b * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1)) * a.x_cs : ((size_t)(b * a.H + gy) * a.W + gx) * a.x_cs;
                const size_t pb2 = ((size_t)(b * a.H + gy) * a.W + gx) * a.x2_cs;
#pragma unroll
                for (int s = 0; s < NK1; ++s) {
                    const int ci = s * 32 + kq * 8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) xf[s][j] = (bf16_t)0.0f;
                    if (inimg) xf[s] = (dual && ci >= a.split_c) ? *reinterpret_cast<const bf16x8*>(xb + pb2 + (ci - a.split_c)) : *reinterpret_cast<const bf16x8*>(xa + pa + ci);
                }
            };
            bf16x8 xcur[NK1], xnext[NK1];
            if (wave < nt0) load_x(wave, xcur);        // issued BEFORE the tile barrier: the loads touch no LDS, and the wait at the barrier (2-2.4 k cycles, c2f_bench) covers their latency
            __syncthreads();                           // weights visible / previous tile's readers of lin, ly0 done
            C2FSTAMP(1);
            for (int t = wave; t < nt0; t += NW) {
                if (t + NW < nt0) load_x(t + NW, xnext);
                const int q = t * 16 + p;
                const int qc = min(q, NP0 - 1);
                const int py = div_small(qc, invPW), px = qc - py * PW;
                const int gy = y0 - 2 + py, gx = x0 - 2 + px;
                const bool inimg = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                f32x4 acc[T1];
#pragma unroll
                for (int c = 0; c < T1; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < NK1; ++s)
#pragma unroll
                    for (int c = 0; c < T1; ++c) {
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w1l + (c * NK1 + s) * 1024);
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xcur[s], acc[c], 0, 0, 0);
                    }
#pragma unroll
                for (int s = 0; s < NK1; ++s) xcur[s] = xnext[s];
                f32x4 o[T1];
#pragma unroll
                for (int c = 0; c < T1; ++c) {
                    f32x4 v = acc[c] + bias_of(2, c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = inimg ? silu<bf16_t>(v[r]) : 0.0f;      // outside the frame: the 3x3 convs' zero padding
                    o[c] = v;
                }
                if (q < NP0) {
                    put_map(lin + (size_t)qc * G::PITCH, o + G::CT);                          // y1: the bottleneck's input
                    const int iy = py - 2, ix = px - 2;
                    const bool interior = (unsigned)iy < (unsigned)a.TH && (unsigned)ix < (unsigned)a.TW && inimg;
                    if (interior) {
                        if (MODE == 3) put_map(ly0 + (size_t)(iy * a.TW + ix) * G::PITCH, o);
                        if (MODE == 1 || a.dump) {
                            const int dst = ((b * a.H + gy) * a.W + gx) * a.cat_cs;
                            put_global(dst, o);
                            put_global(dst + C, o + G::CT);
                        }
                    }
                }
            }
            C2FSTAMP(2);                               // cv1 loop
            __syncthreads();                           // y1 patch complete
            C2FSTAMP(3);
        }

        // ---- conv A: patch -> intermediate map in LDS ----------------------------------------------------------------
        for (int t = wave; t < ntA; t += NW) {
            const int q = t * 16 + p;
            const int qc = min(q, NPA - 1);
            const int my = div_small(qc, invMW), mx = qc - my * MW;
            f32x4 acc[G::CT];
            pair_taps<C>(lin, lwA, (my * PW + mx) * G::PITCH + kq * G::FRAGB, PW * G::PITCH, lane, acc);
            const int gy = y0 - 1 + my, gx = x0 - 1 + mx;
            const bool inimg = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            f32x4 o[G::CT];
#pragma unroll
            for (int c = 0; c < G::CT; ++c) {
                f32x4 v = acc[c] + bias_of(0, c);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = inimg ? silu<bf16_t>(v[r]) : 0.0f;
                o[c] = v;
            }
            if (q < NPA) put_map(lmid + (size_t)(my * MW + mx) * G::PITCH, o);
        }
        C2FSTAMP(4);                                   // conv A loop
        __syncthreads();                               // intermediate map complete
        C2FSTAMP(5);

        // ---- conv B: intermediate -> y (+ shortcut from the patch) ----------------------------------------------------
        for (int t = wave; t < ntB; t += NW) {
            const int q = t * 16 + p;
            const int qc = min(q, NPB - 1);
            const int oy = div_small(qc, invTW), ox = qc - oy * a.TW;
            f32x4 acc[G::CT];
            pair_taps<C>(lmid, lwB, (oy * MW + ox) * G::PITCH + kq * G::FRAGB, MW * G::PITCH, lane, acc);
            f32x4 o[G::CT];
#pragma unroll
            for (int c = 0; c < G::CT; ++c) {
                f32x4 v = acc[c] + bias_of(1, c);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu<bf16_t>(v[r]);
                o[c] = v;
            }
            const int gy = y0 + oy, gx = x0 + ox;
            if (q < NPB) {
                if (a.res) {
                    const unsigned char* xs = lin + ((oy + 2) * PW + ox + 2) * G::PITCH;
                    if (C == 16) o[0] += load4(reinterpret_cast<const bf16_t*>(xs) + kq * 4);
                    else { f32x4 ra, rb; load8(reinterpret_cast<const bf16_t*>(xs) + kq * 8, ra, rb); o[0] += ra; o[1 % G::CT] += rb; }
                }
                if (BACK) put_map(ly2 + (size_t)qc * G::PITCH, o);
                if ((!BACK || a.dump) && gy < a.H && gx < a.W)
                    put_global(((b * a.H + gy) * a.W + gx) * a.cat_cs + a.pair_out_co, o);
            }
        }

        C2FSTAMP(6);                                   // conv B loop
        if (BACK) {
            __syncthreads();                           // y map complete
            C2FSTAMP(7);
            // ---- cv2 over the concat [from HBM: channels below the bottleneck's input | patch interior | y] -> out -----------
            // k-step size = C (one source map per k-step); MODE 3: y0 comes from its LDS map instead of HBM
            const unsigned char* w2l = lw2 + lane * G::FRAGB;
            for (int t = wave; t < ntB; t += NW) {
                const int q = t * 16 + p;
                const int qc = min(q, NPB - 1);
                const int oy = div_small(qc, invTW), ox = qc - oy * a.TW;
                const int gy = min(y0 + oy, a.H - 1), gx = min(x0 + ox, a.W - 1);
                const bf16_t* gp = cat + ((size_t)(b * a.H + gy) * a.W + gx) * a.cat_cs + kq * (G::FRAGB / 2);
                // the k-steps' fragments, in concat order, are all fetched before the MFMAs (MODE 2: the first two from HBM)
                constexpr int NS = MODE == 2 ? 4 : 3;
                F xs[NS];
                if (MODE == 2) {
                    xs[0] = *reinterpret_cast<const F*>(gp);
                    xs[1] = *reinterpret_cast<const F*>(gp + C);
                } else {
                    xs[0] = *reinterpret_cast<const F*>(ly0 + (size_t)qc * G::PITCH + kq * G::FRAGB);
                }
                xs[NS - 2] = *reinterpret_cast<const F*>(lin + (size_t)((oy + 2) * PW + ox + 2) * G::PITCH + kq * G::FRAGB);
                xs[NS - 1] = *reinterpret_cast<const F*>(ly2 + (size_t)qc * G::PITCH + kq * G::FRAGB);
                f32x4 acc[T2MAX];
#pragma unroll
                for (int c = 0; c < T2MAX; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int c = 0; c < T2MAX; ++c)
                        if (c < T2) acc[c] = G::mma(*reinterpret_cast<const F*>(w2l + (c * NS + s) * G::WTILE), xs[s], acc[c]);
                if (q < NPB && y0 + oy < a.H && x0 + ox < a.W) {
                    bf16_t* dst = static_cast<bf16_t*>(a.out) + ((size_t)(b * a.H + gy) * a.W + gx) * a.out_cs + a.out_co;
#pragma unroll
                    for (int g2 = 0; g2 < T2MAX / 2; ++g2) {
                        if (2 * g2 >= T2) break;
                        f32x4 lo = acc[2 * g2] + bias_of(3, 2 * g2), hi = acc[2 * g2 + 1] + bias_of(3, 2 * g2 + 1);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { lo[r] = silu<bf16_t>(lo[r]); hi[r] = silu<bf16_t>(hi[r]); }
                        store8(dst + g2 * 32 + kq * 8, lo, hi);
                    }
                }
            }
        }
        C2FSTAMP(8);                                   // cv2 loop
        if (tnext >= a.total_tiles) break;
        tl = tnext;
        if (!FRONT) __syncthreads();                   // all reads of the patch / maps done before stage_store overwrites them (FRONT: barrier at loop top)
    }
#ifdef ZLY_C2F_DIAG
    if (lane == 0 && g_c2f_diag) {
        unsigned long long* o = g_c2f_diag + ((size_t)blockIdx.x * NW + wave) * 16;
        for (int i = 0; i < 9; ++i) o[i] = dsum[i];
        o[9] = __builtin_amdgcn_s_memtime() - dstart;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// waves per workgroup: c = 32 keeps a conv's 18 weight fragments in registers across its pixel-tile loop (the compiler hoists the
// loop-invariant LDS reads), which needs the 256-VGPR budget of 2 waves per SIMD; c = 16 needs 18 registers for that and runs 16 waves
static constexpr int PAIR_NW16 = 16, PAIR_NW32 = 8, PAIR_NW64 = 8, PAIR_NLD = 4, PAIR_NLD64 = 6;
static int pair_nw(int c) { return c == 16 ? PAIR_NW16 : PAIR_NW32; }
static int pair_nld(int c) { return c == 64 ? PAIR_NLD64 : PAIR_NLD; }
static constexpr int PAIR_LDS_MAX = 160 * 1024;

static int pair_pitch(int c) { return c == 16 ? PairGeom<16>::PITCH : c == 32 ? PairGeom<32>::PITCH : PairGeom<64>::PITCH; }
// LDS bytes of the weight buffer(s): both convs resident for c = 16 / 32, one shared buffer for c = 64
static int pair_wbytes_total(int c) { return c == 16 ? 2 * 9 * PairGeom<16>::WTILE : c == 32 ? 2 * 9 * 2 * PairGeom<32>::WTILE : 9 * 2 * 4 * PairGeom<64>::WTILE; }
static size_t pair_lds_bytes(int c, int th, int tw)
{
    const size_t pitch = (size_t)pair_pitch(c);
    return (size_t)pair_wbytes_total(c) + ((size_t)(th + 4) * (tw + 4) * pitch + 15) / 16 * 16 + (size_t)(th + 2) * (tw + 2) * pitch;
}

// Tile shape for an H x W map and n frames: minimise (rounds of tiles over the 256 CUs) x (16-pixel tile rounds of the
// two convs over the workgroup's waves + a fixed per-tile cost), subject to the LDS budget and the staging registers.
bool pair_plan(int c, int n, int H, int W, PairPlan* plan)
{
    if (c != 16 && c != 32 && c != 64) return false;
    const int ncu = num_cus(), nw = pair_nw(c);
    double best = 1e30;
    for (int th = 4; th <= 32; ++th) {
        for (int tw = 8; tw <= 64; ++tw) {
            if (pair_lds_bytes(c, th, tw) > (size_t)PAIR_LDS_MAX) continue;
            if ((th + 4) * (tw + 4) * (c / 8) > nw * 64 * pair_nld(c)) continue;
            const int tx = (W + tw - 1) / tw, ty = (H + th - 1) / th;
            const long tiles = (long)n * tx * ty;
            const long rounds = (tiles + ncu - 1) / ncu;
            const int ntA = ((th + 2) * (tw + 2) + 15) / 16, ntB = (th * tw + 15) / 16;
            const double per_tile = (double)((ntA + nw - 1) / nw + (ntB + nw - 1) / nw) + 1.5 * 16 / nw;
            const double cost = (double)rounds * per_tile;
            if (cost < best) { best = cost; plan->th = th; plan->tw = tw; plan->tiles_x = tx; plan->tiles_y = ty; plan->total_tiles = (int)tiles; }
        }
    }
    if (best >= 1e30) return false;
    plan->grid = plan->total_tiles < ncu ? plan->total_tiles : ncu;
    plan->lds_bytes = (int)pair_lds_bytes(c, plan->th, plan->tw);
    return true;
}

hipError_t pair_init()
{
    hipError_t r = hipFuncSetAttribute((const void*)bottleneck_pair_kernel<16, PAIR_NW16, PAIR_NLD>, hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_MAX);
    if (r != hipSuccess) return r;
    r = hipFuncSetAttribute((const void*)bottleneck_pair_kernel<32, PAIR_NW32, PAIR_NLD>, hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_MAX);
    if (r != hipSuccess) return r;
    return hipFuncSetAttribute((const void*)bottleneck_pair_kernel<64, PAIR_NW64, PAIR_NLD64>, hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_MAX);
}

// ---- fused C2f kernel -----------------------------------------------------------------------------------------------
static constexpr int C2F_NW16 = 16, C2F_NW32 = 8, C2F_NLD = 4;
// waves per workgroup of the 32-channel kernel: 16 (128 VGPRs, biases in LDS: four waves per SIMD hide the fragment-read and epilogue latencies
// behind each other -- neither the vector nor the matrix pipe was half busy with two) except for the front half, which spills at 128 and keeps 8
// (read where the plan is made -- c2f_plan, cached per engine and shape -- and carried in C2fPlan::nw: as a process-static it could not be switched by a test)
static int c2f_nw32(int mode)
{
    const char* fv = getenv("ZLY_C2F32_NW");                                                   // tuning / tests: 8 or 16 for every mode
    const int forced = fv ? atoi(fv) : 0;
    if (forced == 8 || forced == 16) return forced;
    return mode == 1 ? C2F_NW32 : 16;          // measured at batch 64: back half 40.7 -> 30.2 us, whole block (model.15) 55.9 -> 44.9 us; the front half spills at 128 VGPRs (30.5 -> 34 us)
}
static size_t c2f_lds_bytes(int c, int mode, int nk1, int nk2, int cout2, int th, int tw)
{
    const size_t pitch = (size_t)pair_pitch(c), wtile = c == 16 ? 512 : 1024, ct = c / 16;
    size_t b = C2F_BIAS_BYTES + 2 * 9 * ct * wtile;
    if (mode & 1) b += (size_t)(2 * c / 16) * nk1 * 1024;
    if (mode & 2) b += (size_t)(cout2 / 16) * nk2 * wtile;
    b += ((size_t)(th + 4) * (tw + 4) * pitch + 15) / 16 * 16 + ((size_t)(th + 2) * (tw + 2) * pitch + 15) / 16 * 16;
    if (mode & 2) b += ((size_t)th * tw * pitch + 15) / 16 * 16;
    if (mode == 3) b += ((size_t)th * tw * pitch + 15) / 16 * 16;
    return b;
}

typedef void (*c2f_fn)(const C2fArgs);
static c2f_fn pick_c2f(int c, int mode, int nk1);

// tile shape: rounds of tiles over the CUs x (16-pixel tile rounds of the phases over the workgroup's waves + a fixed cost), LDS budget
bool c2f_plan(int c, int mode, int nk1, int nk2, int cout2, int n, int H, int W, C2fPlan* plan)
{
    if (c == 64 && mode >= 1 && mode <= 3) return c2f64_plan(mode, nk1, nk2, cout2, n, H, W, plan);
    if ((c != 16 && c != 32) || mode < 1 || mode > 3 || !pick_c2f(c, mode, nk1)) return false;
    if ((mode & 2) && nk2 != (mode == 2 ? 4 : 3)) return false;          // concat of 3 (one bottleneck) or 4 (back half of two) sources
    const int ncu = num_cus(), nw = c == 16 ? C2F_NW16 : c2f_nw32(mode);
    double best = 1e30;
    for (int th = 4; th <= 32; ++th) {
        for (int tw = 8; tw <= 64; ++tw) {
            const size_t lds_cap = getenv("ZLY_C2F_LDS_KB") ? (size_t)atoi(getenv("ZLY_C2F_LDS_KB")) * 1024 : (size_t)PAIR_LDS_MAX;    // tuning aid
            if (c2f_lds_bytes(c, mode, nk1, nk2, cout2, th, tw) > lds_cap) continue;
            if (const char* ft = getenv("ZLY_C2F_TILE")) { int fth = 0, ftw = 0; if (sscanf(ft, "%d,%d", &fth, &ftw) == 2 && (fth != th || ftw != tw)) continue; }    // tuning aid: only this shape
            if (!(mode & 1) && (th + 4) * (tw + 4) * (c / 8) > nw * 64 * C2F_NLD) continue;
            const int tx = (W + tw - 1) / tw, ty = (H + th - 1) / th;
            const long tiles = (long)n * tx * ty;
            const long rounds = (tiles + ncu - 1) / ncu;
            const int nt0 = ((th + 4) * (tw + 4) + 15) / 16, ntA = ((th + 2) * (tw + 2) + 15) / 16, ntB = (th * tw + 15) / 16;
            const int ct = c / 16;
            // phase weights in units of one 3x3 conv tile round (9 x CT MFMAs + CT epilogue tiles)
            const double w_cv1 = (double)(2 * ct * nk1 + 2.0 * 2 * ct) / (9.0 * ct + 2.0 * ct);
            const double w_cv2 = (double)((cout2 / 16) * nk2 + 2.0 * (cout2 / 16)) / (9.0 * ct + 2.0 * ct);
            double per_tile = (double)((ntA + nw - 1) / nw + (ntB + nw - 1) / nw) + 1.5 * 16 / nw;
            if (mode & 1) per_tile += w_cv1 * ((nt0 + nw - 1) / nw);
            if (mode & 2) per_tile += w_cv2 * ((ntB + nw - 1) / nw);
            const double cost = (double)rounds * per_tile;
            if (cost < best) { best = cost; plan->th = th; plan->tw = tw; plan->tiles_x = tx; plan->tiles_y = ty; plan->total_tiles = (int)tiles; }
        }
    }
    if (best >= 1e30) return false;
    plan->grid = plan->total_tiles < ncu ? plan->total_tiles : ncu;
    plan->lds_bytes = (int)c2f_lds_bytes(c, mode, nk1, nk2, cout2, plan->th, plan->tw);
    plan->nw = nw;
    return true;
}

// variants built: cv1 with 1, 2 or 6 k-steps of 32 input channels (YOLOv8n: model.2 32 ch, model.4 64 ch, model.15 192 ch); the back half
// (mode 2) has no cv1
template <int C, int NW> static c2f_fn pick_c2f_c(int mode, int nk1)
{
    if (mode == 2) return c2f_kernel<C, 2, NW, C2F_NLD, 1>;
    if (mode == 1) return nk1 == 1 ? c2f_kernel<C, 1, NW, C2F_NLD, 1> : nk1 == 2 ? c2f_kernel<C, 1, NW, C2F_NLD, 2> : nk1 == 6 ? c2f_kernel<C, 1, NW, C2F_NLD, 6> : nullptr;
    return nk1 == 1 ? c2f_kernel<C, 3, NW, C2F_NLD, 1> : nk1 == 2 ? c2f_kernel<C, 3, NW, C2F_NLD, 2> : nk1 == 6 ? c2f_kernel<C, 3, NW, C2F_NLD, 6> : nullptr;
}
static c2f_fn pick_c2f(int c, int mode, int nk1, int nw32)
{
    return c == 16 ? pick_c2f_c<16, C2F_NW16>(mode, nk1) : nw32 == 16 ? pick_c2f_c<32, 16>(mode, nk1) : pick_c2f_c<32, C2F_NW32>(mode, nk1);
}
static c2f_fn pick_c2f(int c, int mode, int nk1) { return pick_c2f(c, mode, nk1, c2f_nw32(mode)); }

hipError_t c2f_init()
{
    static const int nk1s[3] = {1, 2, 6};
    for (int c = 16; c <= 32; c += 16)
        for (int mode = 1; mode <= 3; ++mode)
            for (int i = 0; i < 3; ++i)
                for (int nw32 = 8; nw32 <= 16; nw32 += 8) {
                    hipError_t r = hipFuncSetAttribute((const void*)pick_c2f(c, mode, nk1s[i], nw32), hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_MAX);
                    if (r != hipSuccess) return r;
                }
    return c2f64_init();
}

hipError_t launch_c2f(int c, int mode, const C2fArgs& a, const C2fPlan& plan, hipStream_t s)
{
    if (c == 64) return launch_c2f64(mode, a, plan, s);
    if ((c != 16 && c != 32) || mode < 1 || mode > 3 || a.TH != plan.th || a.TW != plan.tw || plan.grid < 1) return hipErrorInvalidValue;
    if (a.cat_cs % 8 || a.pair_in_co % c || a.pair_out_co % c || a.out_cs % 8 || a.out_co % 8) return hipErrorInvalidValue;
    if ((mode & 1) && (a.x_cs % 8 || a.x_co % 8 || (a.x2 && (a.x2_cs % 8 || a.x2_co % 8 || a.split_c % 32 || (a.H & 1) || (a.W & 1))))) return hipErrorInvalidValue;
    if ((mode & 2) && (a.Cout2 % 32 || a.Cout2 > 64)) return hipErrorInvalidValue;
    const int nw = c == 16 ? C2F_NW16 : plan.nw;
    if (c == 32 && nw != 8 && nw != 16) return hipErrorInvalidValue;
    c2f_fn fn = pick_c2f(c, mode, a.nk1, nw);
    if (!fn || (mode == 2 && a.pair_in_co != 2 * c)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fn, dim3((unsigned)plan.grid), dim3(nw * 64), (size_t)plan.lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_pair(int c, const PairArgs& a, const PairPlan& plan, hipStream_t s)
{
    if ((c != 16 && c != 32 && c != 64) || a.TH != plan.th || a.TW != plan.tw || plan.grid < 1) return hipErrorInvalidValue;
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 8 || a.out_co % 8) return hipErrorInvalidValue;       // 16-byte channel groups
    if (c == 16) hipLaunchKernelGGL((bottleneck_pair_kernel<16, PAIR_NW16, PAIR_NLD>), dim3((unsigned)plan.grid), dim3(PAIR_NW16 * 64), (size_t)plan.lds_bytes, s, a);
    else if (c == 32) hipLaunchKernelGGL((bottleneck_pair_kernel<32, PAIR_NW32, PAIR_NLD>), dim3((unsigned)plan.grid), dim3(PAIR_NW32 * 64), (size_t)plan.lds_bytes, s, a);
    else         hipLaunchKernelGGL((bottleneck_pair_kernel<64, PAIR_NW64, PAIR_NLD64>), dim3((unsigned)plan.grid), dim3(PAIR_NW64 * 64), (size_t)plan.lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace zly
