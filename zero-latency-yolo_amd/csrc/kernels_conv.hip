// kernels_conv.hip -- implicit-GEMM convolution on gfx950 matrix cores (MFMA), NHWC.
//
// Replaces the conv/Gemm nodes that the reference executes inside Ort::Session::Run
// (reference src/inference/onnx_engine.cpp:578-585; ORT 1.8.1 CPU EP, not in the reference tree).
//
// GEMM view:  D[cout][pixel] = sum_k  Wt[cout][k] * X[k][pixel],   k = (ky, kx, ci)
//   * the WEIGHTS are the MFMA "A" operand (rows = output channels), the ACTIVATIONS the "B" operand
//     (columns = output pixels).  With v_mfma_f32_16x16x32_bf16 a lane then ends up holding 4
//     CONSECUTIVE output channels of one pixel, which is exactly one 8-byte (bf16) / 16-byte (fp32)
//     NHWC store -- the transposed assignment would scatter 2-byte stores.
//   * NHWC makes the 8 k-values a lane needs (8 consecutive input channels of one tap) one 16-byte
//     load, so fragments come straight from global memory: no im2col buffer, no LDS round trip.
//   * weights are pre-tiled on the host as [cout/16][k/KSTEP][lane = kq*16 + row][EPL], i.e. each 1 KiB
//     tile is stored in MFMA lane order: one weight fragment for the whole wave is one contiguous
//     1 KiB read from global memory and a conflict-free ds_read_b128 from LDS (weights.cpp: repack_conv).
//   * bias + SiLU + optional residual add + write-at-channel-offset (free Concat / split) are fused
//     into the epilogue.
// fp32 mode uses v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) with a permuted k order inside each
// 16-wide k-step so that both operands still arrive as 16-byte loads.
#include "zly_internal.h"
#include "conv_device.h"
#include <type_traits>
#include <algorithm>

namespace zly {

// Shared epilogue of the conv kernels: bias + SiLU + residual, then NHWC stores.  Tiles c, c+1 of a pair that
// sit in the same wave are written with ONE 16-byte store per lane (8 consecutive channels); 8-byte-per-lane
// stores made the epilogue as expensive as the whole MFMA loop (store-issue bound, tools/diag_lds.hip).
// per-lane bias of the CT channel tiles starting at tile0, fetched ONCE per wave before the main loop: a global
// load inside the epilogue put an exposed L2 round trip (~1-2k cycles under load) at the end of every tile
template <int CT>
__device__ __forceinline__ void load_bias(const ConvArgs& a, int tile0, int kq, f32x4 (&bias)[CT])
{
    const int ntiles = (a.Cout + 15) >> 4;
    const int paired = (ntiles >> 1) << 1;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int ch = tile_channel(tile0 + c, kq, paired);
        bias[c] = ch < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <typename T, int CT>
__device__ __forceinline__ void epilogue_px(const ConvArgs& a, f32x4 (&v)[CT], const f32x4 (&bias)[CT], int tile0, int kq, int m)
{
    const int ntiles = (a.Cout + 15) >> 4;
    const int paired = (ntiles >> 1) << 1;
    f32x4 o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        f32x4 x = v[c] + bias[c];
        if (a.act) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = silu<T>(x[r]);
        }
        o[c] = x;
    }
    const bool even0 = (tile0 & 1) == 0;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int tile = tile0 + c;
        const int ch = tile_channel(tile, kq, paired);
        if (ch >= a.Cout) continue;
        const bool pair_here = even0 && (c % 2 == 0) && (c + 1 < CT) && (tile + 1 < paired);
        if (pair_here) {
            f32x4 lo = o[c], hi = o[c + 1 < CT ? c + 1 : c];
            if (a.res) {
                f32x4 ra, rb;
                load8(static_cast<const T*>(a.res) + (m * a.res_cs + a.res_co + ch), ra, rb);
                lo += ra; hi += rb;
            }
            if (a.out_f32) store8(static_cast<float*>(a.out) + (m * a.out_cs + a.out_co + ch), lo, hi);
            else           store8(static_cast<T*>(a.out) + (m * a.out_cs + a.out_co + ch), lo, hi);
        } else if (!(even0 && (c % 2 == 1) && (tile < paired))) {      // second tile of a pair already written above
            f32x4 x = o[c];
            if (a.res) x += load4(static_cast<const T*>(a.res) + (m * a.res_cs + a.res_co + ch));
            if (a.out_f32) store4(static_cast<float*>(a.out) + (m * a.out_cs + a.out_co + ch), x);
            else           store4(static_cast<T*>(a.out) + (m * a.out_cs + a.out_co + ch), x);
        }
    }
}

// Epilogue of the LDS-tiled kernel (bf16 NHWC output): same arithmetic and channel pairing as epilogue_px, but addresses
// are 32-bit byte offsets into buffer resources (no 64-bit multiply-add per store; out-of-range is the hardware's check)
// and the stores carry their offset in the VGPR operand (see the store-hazard note in conv1x1_stream_kernel).
// The residual fragments epilogue_px_buf will add, requested EARLY (the split-K form asks for them before its k-loop: loaded inside the epilogue they were
// one more exposed global round trip at the end of the five shortcut convs of a batch-1 step).  Same tiles, same offsets, same widths as the epilogue's loads.
template <int CT>
__device__ __forceinline__ void prefetch_res_buf(const ConvArgs& a, __amdgpu_buffer_rsrc_t rres, int tile0, int kq, int m, u32x4 (&rq)[CT])
{
    const int ntiles = (a.Cout + 15) >> 4;
    const int paired = (ntiles >> 1) << 1;
    const int rb = (m * a.res_cs + a.res_co) * 2;
    const bool even0 = (tile0 & 1) == 0;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        rq[c] = u32x4{0u, 0u, 0u, 0u};
        const int tile = tile0 + c;
        const int ch = tile_channel(tile, kq, paired);
        if (ch >= a.Cout) continue;
        const bool pair_here = even0 && (c % 2 == 0) && (c + 1 < CT) && (tile + 1 < paired);
        if (pair_here) rq[c] = __builtin_amdgcn_raw_buffer_load_b128(rres, rb + ch * 2, 0, 0);
        else if (!(even0 && (c % 2 == 1) && (tile < paired))) {
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
            const u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(rres, rb + ch * 2, 0, 0);
            rq[c][0] = r2[0]; rq[c][1] = r2[1];
        }
    }
}

template <int CT, bool PRE = false>
__device__ __forceinline__ void epilogue_px_buf(const ConvArgs& a, __amdgpu_buffer_rsrc_t rout, __amdgpu_buffer_rsrc_t rres,
                                                f32x4 (&v)[CT], const f32x4 (&bias)[CT], int tile0, int kq, int m, const u32x4* rq = nullptr)
{
    const int ntiles = (a.Cout + 15) >> 4;
    const int paired = (ntiles >> 1) << 1;
    const int ob = (m * a.out_cs + a.out_co) * 2;
    const int rb = (m * a.res_cs + a.res_co) * 2;
    f32x4 o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        f32x4 x = v[c] + bias[c];
        if (a.act) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = silu<bf16_t>(x[r]);
        }
        o[c] = x;
    }
    const bool even0 = (tile0 & 1) == 0;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int tile = tile0 + c;
        const int ch = tile_channel(tile, kq, paired);
        if (ch >= a.Cout) continue;
        const bool pair_here = even0 && (c % 2 == 0) && (c + 1 < CT) && (tile + 1 < paired);
        if (pair_here) {
            f32x4 lo = o[c], hi = o[c + 1 < CT ? c + 1 : c];
            if (a.res) {
                const bf16x8 r = __builtin_bit_cast(bf16x8, PRE ? rq[c] : __builtin_amdgcn_raw_buffer_load_b128(rres, rb + ch * 2, 0, 0));
                lo += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
                hi += f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
            }
            const bf16x8 w = to_bf16x8(lo, hi);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, w), rout, ob + ch * 2, 0, 0);
        } else if (!(even0 && (c % 2 == 1) && (tile < paired))) {      // second tile of a pair already written above
            f32x4 x = o[c];
            if (a.res) {
                typedef __attribute__((ext_vector_type(2))) unsigned int u32x2r;
                const bf16x4 r = __builtin_bit_cast(bf16x4, PRE ? u32x2r{rq[c][0], rq[c][1]} : __builtin_amdgcn_raw_buffer_load_b64(rres, rb + ch * 2, 0, 0));
                x += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
            }
            const bf16x4 w = to_bf16x4(x);
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, w), rout, ob + ch * 2, 0, 0);
        }
    }
}

// MODE 0: 1x1 conv.  MODE 1: 3x3, Cin % KSTEP == 0 (a k-step never straddles a tap; tap/ci advance
// incrementally).  MODE 2: 3x3, any Cin % EPL == 0 (tap = k / Cin by reciprocal multiply).
// One wave owns PT 16-pixel tiles x CT 16-channel tiles.
// KSPLIT == 1: a 256-thread workgroup is 4 waves on consecutive pixel tiles of the same channel block
//              (they share the weight fragments through L1); each wave walks the whole K.
// KSPLIT == 4: the 4 waves share ONE set of pixel tiles and each walks a quarter of K; partial sums are
//              reduced through LDS.  Used when M is small (batch 1, deep layers): 4x the workgroups and a
//              4x shorter dependent load->MFMA chain, which is what bounds those launches.
// The k-loop is software pipelined by hand: fragments of step s+1 are requested before the MFMAs of
// step s issue (two named register sets, statically indexed).
#ifdef ZLY_IGEMM_DIAG
__device__ unsigned long long* g_igemm_diag = nullptr;           // diagnostic build only (tools/igemm_bench.hip): per-wave s_memtime stamps of the split-K form
#define IGSTAMP(k) do { if (KSPLIT > 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dst_[k]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define IGSTAMP(k) do { } while (0)
#endif

template <typename T, int MODE, int CT, int PT, int KSPLIT>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a)
{
#ifdef ZLY_IGEMM_DIAG
    unsigned long long dst_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    IGSTAMP(0);                                                // wave start
    typedef typename Frag<T>::type F;
    constexpr int EPL = Frag<T>::EPL;
    constexpr int KSTEP = Frag<T>::KSTEP;
    constexpr int WTILE = 16 * KSTEP;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15;       // pixel within a 16-pixel tile (MFMA B column / D column)
    const int kq = lane >> 4;      // which 8-wide (4-wide for fp32) k group this lane feeds
    int m_base, s_begin, s_end;
    if (KSPLIT == 1) {
        m_base = (blockIdx.x * 4 + wave) * (PT * 16);
        s_begin = 0; s_end = a.nk;
        if (m_base >= a.M) return;     // wave-uniform; this variant has no barriers
    } else {
        m_base = blockIdx.x * (PT * 16);
        const int per = (a.nk + KSPLIT - 1) / KSPLIT;
        s_begin = wave * per;
        s_end = min(a.nk, s_begin + per);
    }

    const T* __restrict__ in = static_cast<const T*>(a.in);

    int boff[PT], iy0[PT], ix0[PT], boff2[PT];
    bool mval[PT];
    const bool dual = MODE == 0 && a.in2 != nullptr;
    const T* __restrict__ in2 = static_cast<const T*>(a.in2);
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const int m = m_base + t * 16 + p;
        mval[t] = m < a.M;
        const int mm = mval[t] ? m : 0;
        if (MODE == 0 && !dual) {
            // plain 1x1 (stride 1, no padding): input pixel = output pixel, no (b, y, x) split needed -- the three
            // integer divisions per pixel tile were a quarter of a wave's instructions in these streaming launches
            iy0[t] = 0; ix0[t] = 0; boff2[t] = 0;
            boff[t] = mm * a.in_cs + a.in_co;
            continue;
        }
        int ox, oy, b;
        if constexpr (KSPLIT > 1) {
            // split-K form (small M by construction): (b, oy, ox) by reciprocal multiplies -- exact for m < 2^20 and maps of < 2^10 pixels a side, which
            // launch_conv checks -- instead of two emulated 32-bit divisions in front of the first load of a ~3 us launch
            const int r = (int)(((float)mm + 0.5f) * a.inv_wo);
            ox = mm - r * a.Wo;
            b = (int)(((float)r + 0.5f) * a.inv_ho);
            oy = r - b * a.Ho;
        } else {
            ox = mm % a.Wo;
            const int r = mm / a.Wo;
            oy = r % a.Ho;
            b = r / a.Ho;
        }
        iy0[t] = oy * a.stride - a.pad;
        ix0[t] = ox * a.stride - a.pad;
        boff[t] = ((b * a.H + iy0[t]) * a.W + ix0[t]) * a.in_cs + a.in_co;
        boff2[t] = 0;
        if (dual) {     // 1x1, stride 1: (iy, ix) = (oy, ox); `in` is the half-size tensor
            boff[t] = ((b * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1)) * a.in_cs + a.in_co;
            boff2[t] = ((b * a.H + oy) * a.W + ox) * a.in2_cs + a.in2_co - a.split_c;
        }
    }

    f32x4 acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* __restrict__ wp = static_cast<const T*>(a.wgt) +
                               (size_t)(blockIdx.y * CT) * a.nk * WTILE + lane * EPL;

    // split-K form: inputs through buffer resources (32-bit byte offsets: launch_conv sends tensors of 2 GiB or more to the KSPLIT = 1 shapes)
    const __amdgpu_buffer_rsrc_t rin1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rin2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in2 ? in2 : in), 0, 0x7fffffff, 0x00020000);
    (void)rin1; (void)rin2;
    // MODE 1 running position of the NEXT step to be loaded (wave-uniform)
    int tap = 0, cbase = 0;
    if (MODE == 1) { const int k0 = s_begin * KSTEP; tap = k0 / a.Cin; cbase = k0 - tap * a.Cin; }
    const float inv_cin = 1.0f / (float)a.Cin;

    auto load_step = [&](int s, F (&wf)[CT], F (&af)[PT]) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
            wf[c] = *reinterpret_cast<const F*>(wp + ((size_t)c * a.nk + s) * WTILE);
        int ci, ky, kx;
        bool kval;
        if (MODE == 0) {
            ci = s * KSTEP + kq * EPL; ky = 0; kx = 0; kval = ci < a.Cin;
        } else if (MODE == 1) {
            ci = cbase + kq * EPL; ky = tap / 3; kx = tap - ky * 3; kval = true;
            cbase += KSTEP;
            if (cbase == a.Cin) { cbase = 0; ++tap; }
        } else {
            const int k = s * KSTEP + kq * EPL;
            const int tp = (int)(((float)k + 0.5f) * inv_cin);      // exact for k < 2^20: frac(k/Cin) >= 1/Cin
            ci = k - tp * a.Cin; ky = tp / 3; kx = tp - ky * 3; kval = tp < 9;
        }
        const int koff = (ky * a.W + kx) * a.in_cs + ci;
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int iy = iy0[t] + ky, ix = ix0[t] + kx;
            const bool ok = MODE == 0 ? (mval[t] && kval) : (mval[t] && kval && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W);
            F z;
#pragma unroll
            for (int j = 0; j < EPL; ++j) z[j] = (T)0.0f;
            if constexpr (KSPLIT > 1) {
                // The latency path: NO branch around the load.  `ok ? *src : zero` became an exec-masked block with the load and a register move
                // inside it, i.e. `s_waitcnt vmcnt(0)` right behind the first k-step's load -- a full global round trip before the other k-steps'
                // loads were even requested, in every one of the ~30 split-K launches of a batch-1 step.  A buffer load takes the mask as an
                // out-of-range offset (the range check returns zeros): straight-line code, every k-step of the ring requested back to back.
                // Which source a k-step reads is wave-uniform (split_c is a multiple of the k-step): a scalar select of the resource.
                const bool src2 = MODE == 0 && dual && s * KSTEP >= a.split_c;
                const int eoff = src2 ? boff2[t] + koff : boff[t] + koff;
                const unsigned boff_bytes = ok ? (unsigned)eoff * (unsigned)sizeof(T) : 0x80000000u;
                af[t] = __builtin_bit_cast(F, __builtin_amdgcn_raw_buffer_load_b128(src2 ? rin2 : rin1, (int)boff_bytes, 0, 0));
                (void)z;
            } else {
                const T* src = (dual && ci >= a.split_c) ? in2 + (long)(boff2[t] + koff) : in + (long)(boff[t] + koff);
                af[t] = ok ? *reinterpret_cast<const F*>(src) : z;
            }
        }
    };
    auto mma_all = [&](const F (&wf)[CT], const F (&af)[PT]) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int t = 0; t < PT; ++t) acc[c][t] = mma_step(wf[c], af[t], acc[c][t]);
    };

    f32x4 biasr[CT];
    load_bias<CT>(a, blockIdx.y * CT, kq, biasr);         // in flight during the k-loop
    // split-K form, bf16 NHWC output with a shortcut: wave 0 (the one that runs the epilogue) asks for the residual now
    constexpr bool RESPRE = KSPLIT > 1 && sizeof(T) == 2;
    u32x4 resq[PT][CT];
    __amdgpu_buffer_rsrc_t rres_early = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.out), 0,
                                                                          (unsigned)((size_t)a.M * (a.res ? a.res_cs : a.out_cs) * sizeof(T)), 0x00020000);
    if constexpr (RESPRE) {
        if (a.res && !a.out_f32 && wave == 0) {
#pragma unroll
            for (int t = 0; t < PT; ++t) {
                const int m = m_base + t * 16 + p;
                prefetch_res_buf<CT>(a, rres_early, blockIdx.y * CT, kq, m < a.M ? m : 0, resq[t]);
            }
        }
    }
    (void)rres_early; (void)resq;
    if constexpr (KSPLIT > 1) {
        // split-K (the latency path): a wave's quarter of K is 1 .. 18 k-steps and its launch lasts as long as this chain -- with two steps in
        // flight that was (steps / 2) dependent L2 round trips (wave lifetime 2.5 us of a 4.8 us launch at batch 1).  A ring of DEPTH register
        // sets, all requested before the first MFMA: one round trip for up to DEPTH steps.  Same MFMA order into the same accumulators: same bits.
        constexpr int DEPTH = CT * PT <= 2 ? 9 : CT * PT == 3 ? 6 : 2;    // 4 / 5 channel tiles (batch-64 launches of the 13 x 13 stage): two sets, as before -- 162 VGPRs cost model.22.cv2.2.1 2 us
        F wfr[DEPTH][CT], afr[DEPTH][PT];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j)
            if (s_begin + j < s_end) load_step(s_begin + j, wfr[j], afr[j]);
        IGSTAMP(1);                                            // index arithmetic done, every load of the ring requested
#ifdef ZLY_IGEMM_DIAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        IGSTAMP(2);                                            // ... and arrived
#endif
        for (int s = s_begin; s < s_end; s += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                if (s + j < s_end) {
                    mma_all(wfr[j], afr[j]);
                    if (s + j + DEPTH < s_end) load_step(s + j + DEPTH, wfr[j], afr[j]);
                }
            }
        }
    } else {
        F wfA[CT], afA[PT], wfB[CT], afB[PT];
        if (s_begin < s_end) load_step(s_begin, wfA, afA);
        for (int s = s_begin; s < s_end; s += 2) {
            if (s + 1 < s_end) load_step(s + 1, wfB, afB);
            mma_all(wfA, afA);
            if (s + 2 < s_end) load_step(s + 2, wfA, afA);
            if (s + 1 < s_end) mma_all(wfB, afB);
        }
    }

    IGSTAMP(3);                                                // MFMAs issued
    if constexpr (KSPLIT > 1) {
        // partial sums of waves 1..3 -> LDS -> wave 0 (each lane only ever touches its own column)
        __shared__ float red[(KSPLIT - 1) * CT * PT * 4 * 64];
        if (wave > 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[(((wave - 1) * CT + c) * PT + t) * 256 + r * 64 + lane] = acc[c][t][r];
        }
        __syncthreads();
        IGSTAMP(4);                                            // partial sums in LDS, barrier passed
#ifdef ZLY_IGEMM_DIAG
        if (wave > 0 && lane == 0 && g_igemm_diag) {
            unsigned long long* o = g_igemm_diag + (((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
            for (int k = 0; k < 5; ++k) o[k] = dst_[k];
        }
#endif
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < KSPLIT - 1; ++w)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[c][t][r] += red[((w * CT + c) * PT + t) * 256 + r * 64 + lane];
    }

    // epilogue (see epilogue_px): per pixel tile, all CT channel tiles of this lane at once
    __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)((size_t)a.M * a.out_cs * (a.out_f32 ? 4 : sizeof(T))), 0x00020000);
    __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.out), 0,
                                                                    (unsigned)((size_t)a.M * (a.res ? a.res_cs : a.out_cs) * sizeof(T)), 0x00020000);
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const int m = m_base + t * 16 + p;
        if (m >= a.M) continue;
        f32x4 v[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) v[c] = acc[c][t];
        if constexpr (sizeof(T) == 2) {
            if (!a.out_f32) {                                                                                      // bf16 NHWC output: 32-bit buffer offsets
                if constexpr (RESPRE) epilogue_px_buf<CT, true>(a, rout, rres, v, biasr, blockIdx.y * CT, kq, m, resq[t]);
                else                  epilogue_px_buf<CT>(a, rout, rres, v, biasr, blockIdx.y * CT, kq, m);
                continue;
            }
        }
        epilogue_px<T, CT>(a, v, biasr, blockIdx.y * CT, kq, m);
    }
#ifdef ZLY_IGEMM_DIAG
    if (KSPLIT > 1) {
        IGSTAMP(5);                                            // epilogue done, stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        IGSTAMP(6);                                            // stores acknowledged
        if (lane == 0 && g_igemm_diag) {
            unsigned long long* o = g_igemm_diag + (((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
            for (int k = 0; k < 7; ++k) o[k] = dst_[k];
        }
    }
#endif
}

// (Fetching the arguments up front -- all of them, or only the ~100 bytes the index arithmetic needs -- was measured SLOWER than the compiler's fetch-at-first-use on
// the batch-1 step: +2.4 % / +1.3 %.  What helps is the ORDER of the fields in ConvArgs: zly_internal.h.)
template <typename T, int MODE, int CT, int PT, int KSPLIT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a)
{
    conv_igemm_body<T, MODE, CT, PT, KSPLIT>(a);
}

// Several independent 3x3 convs (same kernel shape: bf16, Cin % 32 == 0, CT channel tiles per wave, one 16-pixel tile per
// workgroup, 4-way split-K) in ONE launch: blockIdx.z picks the conv.  The latency path (batch <= 4) is a chain of ~40 launches
// of a few microseconds each; the Detect head's three pyramid levels are independent of each other, so their three stem convs go
// in one launch and the six second convs of their box / class branches in another (9 launches -> 2).
template <int CT>
__global__ __launch_bounds__(256) void conv_igemm_multi_kernel(const ConvArgsMulti m)
{
    const ConvArgs& a = m.a[blockIdx.z];
    if (((int)blockIdx.y * CT * 16 >= a.cout_pad) | ((int)blockIdx.x * 16 >= a.M)) return;   // block-uniform: beyond this conv's extent
    conv_igemm_body<bf16_t, 1, CT, 1, 4>(a);
}

hipError_t launch_conv_multi(const ConvArgsMulti& m, int ct, hipStream_t s)
{
    if (m.n < 1 || m.n > 6 || (ct != 2 && ct != 3)) return hipErrorInvalidValue;
    int gx = 0, gy = 0;
    for (int i = 0; i < m.n; ++i) {
        const ConvArgs& a = m.a[i];
        if (a.Cin % 32 || a.pad != 1 || a.in2 || a.out_f32 || a.cout_pad % (16 * ct)) return hipErrorInvalidValue;
        if ((size_t)(a.M / (a.Ho * a.Wo)) * a.H * a.W * (size_t)a.in_cs * 2 >= ((size_t)1 << 31)) return hipErrorInvalidValue;     // 32-bit buffer offsets (batch <= 4 launches: far below)
        gx = gx > (a.M + 15) / 16 ? gx : (a.M + 15) / 16;
        gy = gy > a.cout_pad / (16 * ct) ? gy : a.cout_pad / (16 * ct);
    }
    ConvArgsMulti mm = m;
    for (int i = 0; i < mm.n; ++i) {
        if (mm.a[i].M >= (1 << 20) || mm.a[i].Wo >= 1024 || mm.a[i].Ho >= 1024) return hipErrorInvalidValue;       // reciprocal index arithmetic of the split-K form
        mm.a[i].inv_wo = 1.0f / (float)mm.a[i].Wo; mm.a[i].inv_ho = 1.0f / (float)mm.a[i].Ho;
    }
    if (ct == 2) hipLaunchKernelGGL(conv_igemm_multi_kernel<2>, dim3(gx, gy, mm.n), dim3(256), 0, s, mm);
    else         hipLaunchKernelGGL(conv_igemm_multi_kernel<3>, dim3(gx, gy, mm.n), dim3(256), 0, s, mm);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolution, streaming variant (bf16, stride 1, single source, Cin <= 32 * NK): the throughput kernel of the
// pointwise layers.  These launches move 20-110 MB through a handful of MFMAs per pixel; with one pixel group per
// wave (kernel above) every wave exposes a full HBM/MALL round trip before its first MFMA and the chip holds too few
// bytes in flight.  Here a wave is persistent over pixel groups (PT x 16 pixels, all of K) and always has the NEXT
// group's activation fragments in flight while it runs the MFMAs and the SiLU epilogue of the current one; the weight
// fragments of its CT channel tiles (CT x NK <= 16) stay in registers for the whole launch; no index divisions.
// ------------------------------------------------------------------------------------------------
template <int CT, int PT, int NK>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(const ConvArgs a, int ngroups)
{
    static_assert(CT % 2 == 0, "channel tiles are stored in pairs (8 consecutive channels per lane)");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, kq = lane >> 4;
    const int gstride = gridDim.x * 4;
    int g = blockIdx.x * 4 + wave;
    if (g >= ngroups) return;

    // Buffer resources over the input view and the output view: addresses are (wave-uniform byte offset of the pixel
    // group) + (per-lane byte offset, computed once) + immediate, both in the VGPR operand -- soffset is NOT part of the
    // hardware range check, so a group offset passed there would let the last, partial group read past the tensor --
    // out-of-range pixels of the last group read zeros and their stores are dropped by the range check: one v_add per
    // access, no bounds VALU.
    const bf16_t* inb = static_cast<const bf16_t*>(a.in) + a.in_co;
    bf16_t* outb = static_cast<bf16_t*>(a.out) + a.out_co + blockIdx.y * (CT * 16);
    const unsigned in_bytes = (unsigned)(((size_t)(a.M - 1) * a.in_cs + min(a.Cin, a.in_cs - a.in_co)) * 2);
    const unsigned out_bytes = (unsigned)(((size_t)(a.M - 1) * a.out_cs + CT * 16) * 2);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(inb), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(outb, 0, out_bytes, 0x00020000);
    int vin[PT], vout[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        vin[t] = ((t * 16 + p) * a.in_cs + kq * 8) * 2;
        vout[t] = ((t * 16 + p) * a.out_cs + kq * 8) * 2;
    }
    const int gin = PT * 16 * a.in_cs * 2, gout = PT * 16 * a.out_cs * 2;       // bytes per pixel group

    bf16x8 w[CT][NK];
    {
        const bf16_t* __restrict__ wp = static_cast<const bf16_t*>(a.wgt) + (size_t)(blockIdx.y * CT) * a.nk * 512 + lane * 8;
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int s = 0; s < NK; ++s) w[c][s] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)c * a.nk + s) * 512);
    }
    // pair-permuted channel tiles (weights.cpp): tile 2j holds channels j*32 + kq*8 .. +3, tile 2j+1 the next four
    f32x4 biasr[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
        biasr[c] = *reinterpret_cast<const f32x4*>(a.bias + blockIdx.y * (CT * 16) + (c >> 1) * 32 + kq * 8 + (c & 1) * 4);

    // K padding (Cin % 32 != 0, e.g. the 48-channel concat of model.2.cv2): the lanes of the last k-step whose channels lie
    // beyond Cin would read the next pixel's first channels; their weights are zero, but 0 * Inf/NaN is not, so they are
    // masked like conv_igemm_kernel does (a per-lane constant: four v_cndmask per fragment of the last k-step)
    const bool kpad_lane = (NK - 1) * 32 + kq * 8 >= a.Cin;
    auto load_group = [&](int grp, u32x4 (&x)[PT][NK]) {
        const int so = grp * gin;
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int s = 0; s < NK; ++s) {
                x[t][s] = __builtin_amdgcn_raw_buffer_load_b128(rin, vin[t] + so + s * 64, 0, 0);
                if (s == NK - 1 && kpad_lane) x[t][s] = u32x4{0u, 0u, 0u, 0u};
            }
    };
    auto compute = [&](int grp, const u32x4 (&x)[PT][NK]) {
        f32x4 acc[CT][PT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int t = 0; t < PT; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NK; ++s)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[c][t] = mma_step(w[c][s], __builtin_bit_cast(bf16x8, x[t][s]), acc[c][t]);
        const int so = grp * gout;
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int c = 0; c < CT; c += 2) {
                f32x4 lo = acc[c][t] + biasr[c], hi = acc[c + 1][t] + biasr[c + 1];
#pragma unroll
                for (int r = 0; r < 4; ++r) { lo[r] = silu<bf16_t>(lo[r]); hi[r] = silu<bf16_t>(hi[r]); }
                const bf16x8 o = to_bf16x8(lo, hi);
                // the group offset goes into the VGPR offset, not soffset: with an SGPR soffset hipcc emits no wait state
                // between a 16-byte buffer store and the next VALU write of its data registers (LLVM's hazard rule
                // exempts that form), and on gfx950 lanes 12-15 of dword 1 were then stored from the overwritten register
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rout, vout[t] + so + (c >> 1) * 64, 0, 0);
            }
    };

    u32x4 xA[PT][NK], xB[PT][NK];
    load_group(g, xA);
    while (true) {
        const int g1 = g + gstride;
        if (g1 < ngroups) load_group(g1, xB);
        compute(g, xA);
        if (g1 >= ngroups) break;
        const int g2 = g1 + gstride;
        if (g2 < ngroups) load_group(g2, xA);
        compute(g1, xB);
        if (g2 >= ngroups) break;
        g = g2;
    }
}

typedef void (*conv_stream_fn)(const ConvArgs, int);
// (CT, PT, NK) shapes built: CT x NK <= 16 weight fragments, PT x NK <= 8 activation fragments per buffer
static conv_stream_fn pick_stream(int ct, int pt, int nk)
{
#define ZLY_STREAM_CASE(C_, P_, N_) if (ct == C_ && pt == P_ && nk == N_) return conv1x1_stream_kernel<C_, P_, N_>
    ZLY_STREAM_CASE(2, 4, 1); ZLY_STREAM_CASE(2, 4, 2);
    ZLY_STREAM_CASE(4, 2, 1); ZLY_STREAM_CASE(4, 2, 2); ZLY_STREAM_CASE(4, 2, 3); ZLY_STREAM_CASE(4, 2, 4);
    ZLY_STREAM_CASE(2, 1, 6); ZLY_STREAM_CASE(2, 1, 8);
    ZLY_STREAM_CASE(4, 1, 6); ZLY_STREAM_CASE(4, 1, 8);          // 64 channels per wave: the input is read twice instead of four times for 128 outputs
#undef ZLY_STREAM_CASE
    return nullptr;
}

// Fragment reads of the tap loop are software-pipelined ZLY_TAPS_DEPTH taps ahead of the MFMAs that consume them
// (statically indexed fragment sets) and pinned with sched_barrier.  Left to itself hipcc issues each tap's
// ds_reads 1-2 instructions before its MFMAs with lgkmcnt(0/1) waits in between.  Measured with the stamped
// build (tools/diag_lds.hip): tap phase 2100 -> 1700 cycles per item at depth 1; depth 2 costs VGPRs (occupancy)
// for no further gain; 0 = compiler schedule.
#ifndef ZLY_TAPS_DEPTH
#define ZLY_TAPS_DEPTH 1
#endif
#ifndef ZLY_LDS_MIN_WAVES
#define ZLY_LDS_MIN_WAVES 2      // waves per SIMD the register allocation must allow (= resident workgroups per CU of this 4-wave kernel)
#endif
#ifndef ZLY_LDS_WDMA
#define ZLY_LDS_WDMA 1           // per-item weight tiles go global -> LDS by LDS-DMA (buffer_load ... lds): no register round trip, no ds_write;
                                 // 186 -> 133 registers for the CT=4 x PT=2 variant (3 resident workgroups per CU), +3 % on the whole step.  0 = through registers
#endif
#ifndef ZLY_LDS_DEPTH
#define ZLY_LDS_DEPTH 1          // items of global loads in flight ahead of the one being computed.  2 (a second register set,
                                 // +48..60 VGPRs) was needed while the kernel ran one workgroup per CU; with two or three resident
                                 // workgroups covering each other it measures the same (tools/diag_lds.hip, -DZLY_LDS_DEPTH=2)
#endif
#ifndef ZLY_TAPS_PIN
#define ZLY_TAPS_PIN 1
#endif
#if ZLY_TAPS_DEPTH > 0
template <int CT, int PT, int S, int PW, int PITCH>
__device__ __forceinline__ void taps_mma(const unsigned char* lpatch, const unsigned char* lw, int lane, int row0, int p, int kq,
                                         f32x4 (&acc)[CT][PT])
{
    constexpr int D = ZLY_TAPS_DEPTH;
    bf16x8 wf[D + 1][CT], af[D + 1][PT];
    const unsigned char* wl = lw + lane * 16;
    const unsigned char* al = lpatch + ((row0 * S) * PW + p * S) * PITCH + kq * 16;
    auto rd = [&](int t, bf16x8 (&w)[CT], bf16x8 (&x)[PT]) {
        const int ky = t / 3, kx = t - ky * 3;
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) w[cc] = *reinterpret_cast<const bf16x8*>(wl + (t * CT + cc) * 1024);
#pragma unroll
        for (int i = 0; i < PT; ++i) x[i] = *reinterpret_cast<const bf16x8*>(al + ((i * S + ky) * PW + kx) * PITCH);
    };
#pragma unroll
    for (int t = 0; t < D; ++t) rd(t, wf[t], af[t]);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t + D < 9) rd(t + D, wf[(t + D) % (D + 1)], af[(t + D) % (D + 1)]);
#if ZLY_TAPS_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int cc = 0; cc < CT; ++cc)
#pragma unroll
            for (int i = 0; i < PT; ++i) acc[cc][i] = mma_step(wf[t % (D + 1)][cc], af[t % (D + 1)][i], acc[cc][i]);
#if ZLY_TAPS_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}
#endif

// ------------------------------------------------------------------------------------------------
// 3x3 convolution, LDS-tiled (bf16, Cin % 32 == 0, pad 1, stride S): the throughput kernel.
//
// The direct kernel above re-reads every input pixel 9x (once per tap) and the whole weight matrix once
// per 16 pixels through L1; at batch 64 that traffic, not MFMA or HBM, bounds it.  Here a workgroup owns
// a TH x 16 output tile of one frame and walks Cin in chunks of 32 channels.  Per chunk it stages
//   * the input patch ((TH-1)S+3) x (15S+3) pixels x 32 ch  -> LDS, pixel pitch 96 B (S=1) / 80 B (S=2):
//     conflict-free for the ds_read_b128 fragment pattern (tests/lds_pitch.py), zero-filled outside the frame
//   * the 9 x CT weight tiles (1 KiB each, already in MFMA lane order)                     -> LDS
// once, and all 9 taps x CT x PT MFMAs of the 4 waves read their fragments from LDS.
// Workgroups are persistent over (tile, chunk) items: the global loads of item i+1 are issued into
// registers before the MFMAs of item i and written to LDS after them (one LDS buffer, so two
// workgroups fit per CU and cover each other's barriers).
// ------------------------------------------------------------------------------------------------

template <int S, int PT> struct LdsGeom {
    static constexpr int TH = 4 * PT, TW = 16;
    static constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
    static constexpr int PITCH = S == 1 ? 96 : 80;
    static constexpr int PATCH_BYTES = (PH * PW * PITCH + 15) / 16 * 16;
};

template <int S, int CT, int PT>
__global__ __launch_bounds__(256, ZLY_LDS_MIN_WAVES) void conv3x3_lds_kernel(const ConvArgs a, int tiles_x, int tiles_per_img, int total_tiles, int wres)
{
    typedef LdsGeom<S, PT> G;
    constexpr int PW = G::PW, PITCH = G::PITCH;
    constexpr int NPU = G::PH * G::PW * 4;            // 16-byte units in the patch chunk
    constexpr int NPU_T = (NPU + 255) / 256;
    constexpr int NWU = 9 * CT * 64;                  // 16-byte units in the weight chunk
    constexpr int NWU_T = (NWU + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lpatch = smem;
    unsigned char* lw = smem + G::PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kq = lane >> 4;
    const int nchunks = a.Cin >> 5;
    const bf16_t* __restrict__ in = static_cast<const bf16_t*>(a.in) + a.in_co;
    const bf16_t* __restrict__ wbase = static_cast<const bf16_t*>(a.wgt) + (size_t)(blockIdx.y * CT) * a.nk * 512;

    // per-thread staging geometry, computed once: unit i of this thread is pixel (upy, upx) of the patch,
    // 16-byte piece q; its source offset relative to the patch origin and its weight offset are constants
    int upy[NPU_T], upx[NPU_T], usrc[NPU_T], uwsrc[NWU_T];
#pragma unroll
    for (int i = 0; i < NPU_T; ++i) {
        const int u = min(tid + i * 256, NPU - 1);
        const int px = u >> 2, q = u & 3;
        upy[i] = px / PW; upx[i] = px - upy[i] * PW;
        usrc[i] = (upy[i] * a.W + upx[i]) * a.in_cs + q * 8;
    }
#pragma unroll
    for (int i = 0; i < NWU_T; ++i) {
        const int u = min(tid + i * 256, NWU - 1);
        const int ti = u >> 6, l = u & 63;                     // ti = tap * CT + ct
        const int t = ti / CT, ct = ti - t * CT;
        uwsrc[i] = (ct * a.nk + t * nchunks) * 512 + l * 8;
    }
    // wres: the weight tiles of ALL chunks of this workgroup's channel block fit in LDS next to the patch (host decision:
    // Cin / 32 x 9 x CT KiB): they are copied once, before the item loop, and items stage the patch only.  Weights were
    // 2/3 of the bytes staged per item (staging costs ~1/64 + 1/79 cycles per byte per CU): with them resident a
    // 64 -> 64 layer runs as 2 channel blocks of CT = 2 that stage 17 KB per item instead of one block staging 54 KB.
    const bool w_once = wres != 0;
    bool w_staged = w_once;
    if (w_once) {
        for (int c = 0; c < nchunks; ++c) {
            const bf16_t* wc = wbase + c * 512;
#pragma unroll
            for (int i = 0; i < NWU_T; ++i) {
                const int u = tid + i * 256;
                if (u < NWU) *reinterpret_cast<u32x4*>(lw + (size_t)c * (NWU * 16) + u * 16) = *reinterpret_cast<const u32x4*>(wc + uwsrc[i]);
            }
        }
    }
    auto stage_load = [&](int tl, int c, u32x4 (&rp)[NPU_T], u32x4 (&rw)[NWU_T]) {
        const int b = tl / tiles_per_img;
        const int r = tl - b * tiles_per_img;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int iy_base = ty * G::TH * S - 1, ix_base = tx * G::TW * S - 1;
        // wave-uniform base of the patch origin (may point before the frame: only dereferenced when in range)
        const bf16_t* inb = in + ((long)b * a.H * a.W + (long)iy_base * a.W + ix_base) * a.in_cs + c * 32;
#pragma unroll
        for (int i = 0; i < NPU_T; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if ((unsigned)(iy_base + upy[i]) < (unsigned)a.H && (unsigned)(ix_base + upx[i]) < (unsigned)a.W)
                v = *reinterpret_cast<const u32x4*>(inb + usrc[i]);
            rp[i] = v;
        }
        if (w_once || ZLY_LDS_WDMA) return;                   // weights are resident / moved by LDS-DMA
        const bf16_t* wc = wbase + c * 512;
#pragma unroll
        for (int i = 0; i < NWU_T; ++i) rw[i] = *reinterpret_cast<const u32x4*>(wc + uwsrc[i]);
    };
    auto stage_store = [&](const u32x4 (&rp)[NPU_T], const u32x4 (&rw)[NWU_T]) {
#pragma unroll
        for (int i = 0; i < NPU_T; ++i) {
            const int u = tid + i * 256;
            if (u < NPU) *reinterpret_cast<u32x4*>(lpatch + (u >> 2) * PITCH + (u & 3) * 16) = rp[i];
        }
        if ((w_once && w_staged) || ZLY_LDS_WDMA) return;
        w_staged = true;
#pragma unroll
        for (int i = 0; i < NWU_T; ++i) {
            const int u = tid + i * 256;
            if (u < NWU) *reinterpret_cast<u32x4*>(lw + u * 16) = rw[i];
        }
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

#if ZLY_LDS_WDMA
    // LDS-DMA of one chunk's weight tiles: tile ti = tap * CT + ct is one contiguous KiB in lane order on both sides; wave w
    // moves tiles w, w+4, ...; completion is the wave's vmcnt, visibility the workgroup barrier that follows
    const __amdgpu_buffer_rsrc_t rwgt = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wbase), 0, (unsigned)(CT * a.nk * 1024), 0x00020000);
    auto dma_weights = [&](int c) {
#pragma unroll
        for (int i = 0; i < (9 * CT + 3) / 4; ++i) {
            const int ti = wave + 4 * i;
            if (ti < 9 * CT) {
                const int t = ti / CT, ct = ti - t * CT;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rwgt, (__attribute__((address_space(3))) void*)(lw + ti * 1024), 16, lane * 16,
                                                         (ct * a.nk + t * nchunks + c) * 1024, 0, 0);
            }
        }
    };
#endif
    f32x4 biasr[CT];
    load_bias<CT>(a, blockIdx.y * CT, kq, biasr);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)((size_t)a.M * a.out_cs * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.out), 0,
                                                                          (unsigned)((size_t)a.M * (a.res ? a.res_cs : a.out_cs) * 2), 0x00020000);
#ifdef ZLY_DIAG
    unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, ditems = 0, dT0 = 0, dT1 = 0;
#define ZSTAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define ZPHASE(k) do { ZSTAMP(dT1); dsum[k] += dT1 - dT0; dT0 = dT1; } while (0)
#endif
    // 9 taps of one staged chunk; the epilogue runs after a tile's last chunk
    bool dma_next_valid = false; int dma_next_c = 0;
    (void)dma_next_valid; (void)dma_next_c;
    auto compute = [&](int tl, int c) {
#if ZLY_TAPS_DEPTH > 0
        taps_mma<CT, PT, S, PW, PITCH>(lpatch, w_once ? lw + (size_t)c * (NWU * 16) : lw, lane, wave * PT, p, kq, acc);
#else
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            bf16x8 wf[CT], af[PT];
#pragma unroll
            for (int cc = 0; cc < CT; ++cc)
                wf[cc] = *reinterpret_cast<const bf16x8*>((w_once ? lw + (size_t)c * (NWU * 16) : lw) + (t * CT + cc) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int row = wave * PT + i;
                af[i] = *reinterpret_cast<const bf16x8*>(lpatch + ((row * S + ky) * PW + p * S + kx) * PITCH + kq * 16);
            }
#pragma unroll
            for (int cc = 0; cc < CT; ++cc)
#pragma unroll
                for (int i = 0; i < PT; ++i) acc[cc][i] = mma_step(wf[cc], af[i], acc[cc][i]);
        }
#endif
#ifdef ZLY_DIAG
        ZPHASE(3);
#endif
#if ZLY_LDS_WDMA
        if (!w_once) {
            __syncthreads();                                   // every wave is done reading this item's weights and patch
            if (dma_next_valid) dma_weights(dma_next_c);       // the next item's weights land during the epilogue / patch store
        }
#endif
        if (c != nchunks - 1) return;
        // epilogue for tile tl (see epilogue_px)
        const int b = tl / tiles_per_img;
        const int r = tl - b * tiles_per_img;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int ox = tx * G::TW + p;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int oy = ty * G::TH + wave * PT + i;
            f32x4 v[CT];
#pragma unroll
            for (int cc = 0; cc < CT; ++cc) { v[cc] = acc[cc][i]; acc[cc][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if (oy >= a.Ho || ox >= a.Wo) continue;
            epilogue_px_buf<CT>(a, rout, rres, v, biasr, blockIdx.y * CT, kq, (b * a.Ho + oy) * a.Wo + ox);
        }
    };

    // item stream of this workgroup: (tile, chunk) with tile = blockIdx.x + k * gridDim.x; two items are
    // always in flight in registers (sets A and B) ahead of the one being computed from LDS
    auto advance = [&](int& tl, int& c) { if (++c == nchunks) { c = 0; tl += gridDim.x; } };
    int t0 = blockIdx.x, c0 = 0;                 // item being computed
    if (t0 >= total_tiles) return;
    int t1 = t0, c1 = c0; advance(t1, c1);       // next item
    int t2 = t1, c2 = c1; advance(t2, c2);       // the one after
#if ZLY_LDS_DEPTH == 1
    u32x4 rpA[NPU_T], rwA[NWU_T];
    stage_load(t0, c0, rpA, rwA);
#else
    u32x4 rpA[NPU_T], rwA[NWU_T], rpB[NPU_T], rwB[NWU_T];
    stage_load(t0, c0, rpA, rwA);
    if (t1 < total_tiles) stage_load(t1, c1, rpB, rwB);
#endif
#ifdef ZLY_DIAG
    // diagnostic build only (tools/diag_lds.hip): per-wave cycle sums of the phases of an item, written to the
    // buffer passed in a.in2 (unused by 3x3 convs): [store+wait, barrier, load issue, taps+epilogue, barrier, items]

    const unsigned long long dstart = __builtin_amdgcn_s_memtime();
    const unsigned long long drt0 = __builtin_amdgcn_s_memrealtime();        // constant 100 MHz: wall clock
    ZSTAMP(dT0);
#else
#define ZPHASE(k) do { } while (0)
#endif
#if ZLY_LDS_DEPTH == 1
#if ZLY_LDS_WDMA
    if (!w_once) dma_weights(c0);
#endif
    while (true) {
        stage_store(rpA, rwA);
#if ZLY_LDS_WDMA
        if (!w_once) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this item's weight DMA (and whatever was issued after it) has landed
#endif
        ZPHASE(0);
        __syncthreads();
        ZPHASE(1);
        if (t1 < total_tiles) stage_load(t1, c1, rpA, rwA);
        ZPHASE(2);
        dma_next_valid = t1 < total_tiles; dma_next_c = c1;
        compute(t0, c0);
        ZPHASE(5);
#if ZLY_LDS_WDMA
        if (w_once) __syncthreads();                           // (the DMA path has its barrier between the taps and the epilogue)
#else
        __syncthreads();
#endif
        ZPHASE(4);
#ifdef ZLY_DIAG
        ++ditems;
#endif
        if (t1 >= total_tiles) break;
        t0 = t1; c0 = c1; advance(t1, c1);
    }
#else
    while (true) {
        // ---- item (t0,c0) from set A ----
        stage_store(rpA, rwA);
        ZPHASE(0);
        __syncthreads();
        ZPHASE(1);
        if (t2 < total_tiles) stage_load(t2, c2, rpA, rwA);
        ZPHASE(2);
        compute(t0, c0);
        ZPHASE(5);
        __syncthreads();
        ZPHASE(4);
#ifdef ZLY_DIAG
        ++ditems;
#endif
        if (t1 >= total_tiles) break;
        // ---- item (t1,c1) from set B ----
        int t3 = t2, c3 = c2; advance(t3, c3);
        stage_store(rpB, rwB);
        ZPHASE(0);
        __syncthreads();
        ZPHASE(1);
        if (t3 < total_tiles) stage_load(t3, c3, rpB, rwB);
        ZPHASE(2);
        compute(t1, c1);
        ZPHASE(5);
        __syncthreads();
        ZPHASE(4);
#ifdef ZLY_DIAG
        ++ditems;
#endif
        if (t2 >= total_tiles) break;
        t0 = t2; c0 = c2; t1 = t3; c1 = c3;
        t2 = t1; c2 = c1; advance(t2, c2);
    }
#endif
#ifdef ZLY_DIAG
    if (lane == 0 && a.in2) {
        unsigned long long* o = (unsigned long long*)a.in2 + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16;
        o[8] = drt0; o[9] = __builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < 5; ++k) o[k] = dsum[k];
        o[5] = ditems; o[6] = __builtin_amdgcn_s_memtime() - dstart; o[7] = dsum[5];
    }
#endif
}

typedef void (*conv_lds_fn)(const ConvArgs, int, int, int, int);

template <int S, int PT>
static conv_lds_fn pick_lds_ct(int ct) {
    switch (ct) {
        case 2: return conv3x3_lds_kernel<S, 2, PT>;
        case 3: return conv3x3_lds_kernel<S, 3, PT>;
        case 4: return conv3x3_lds_kernel<S, 4, PT>;
        case 5: return conv3x3_lds_kernel<S, 5, PT>;
    }
    return nullptr;
}
static conv_lds_fn pick_lds(int stride, int pt, int ct) {
    if (stride == 1) return pt == 1 ? pick_lds_ct<1, 1>(ct) : pt == 2 ? pick_lds_ct<1, 2>(ct) : pick_lds_ct<1, 4>(ct);
    return pt == 1 ? pick_lds_ct<2, 1>(ct) : pick_lds_ct<2, 2>(ct);
}
static size_t lds_bytes(int stride, int pt, int ct, int wchunks = 1) {
    size_t patch = 0;
    if (stride == 1) patch = pt == 1 ? LdsGeom<1, 1>::PATCH_BYTES : pt == 2 ? LdsGeom<1, 2>::PATCH_BYTES : LdsGeom<1, 4>::PATCH_BYTES;
    else             patch = pt == 1 ? LdsGeom<2, 1>::PATCH_BYTES : LdsGeom<2, 2>::PATCH_BYTES;
    return patch + (size_t)wchunks * 9 * ct * 1024;
}

// ------------------------------------------------------------------------------------------------
// 3x3 convolution, WEIGHT-STATIONARY PER WAVE (bf16, stride S = 1 or 2, pad 1, Cin = 32 * NKS / 9): the kernel of the FLOP-dense layers -- the
// Detect branches at P3 / P4, the 64-channel bottlenecks at 26 x 26 and (S = 2) the down-sampling convs with 64 input channels.
//
// What the stamped builds showed (profiles/r03_lds_kernel_phase_stamps.txt, r03_ps_kernel_phase_stamps.txt): conv3x3_lds_kernel and a
// first pixel-stationary variant with a weight ring in LDS both ran the P3 stem at ~560 TFLOP/s although their inner loop alone does
// 1770 (tools/mfma_bench.hip): per tile a wave spent 30 k of 50 k cycles issuing weight DMAs (~100 cycles per KiB), waiting for them,
// in one barrier per k-step and in the epilogue.  The weights are the problem, so here they never move:
//   * the output channels are split over the waves of a workgroup: a wave owns TPW = 2 of the 16-channel tiles (8 consecutive channels
//     per lane with the pair-permuted rows) and keeps their weight fragments for ALL NKS k-steps in registers (2 x 18 x 4 = 144 VGPRs
//     for Cin = 64), loaded once per workgroup;
//   * a workgroup owns a TH x TW pixel tile of one frame (13 x 26 divides the 52 x 52 and 26 x 26 maps); its input patch goes to LDS once
//     (pixel pitch Cin*2 + 32 B: conflict-free ds_read_b128); the pixels are linearised into 16-pixel column tiles and each wave
//     streams the column tiles of its pixel group past its weights: per column tile NKS pixel fragments from LDS, TPW * NKS MFMAs,
//     bias + SiLU + ONE 16-byte NHWC store per lane -- no barrier, no wait on another wave, anywhere in the loop;
//   * a workgroup = 4 waves = NWC channel groups x NWP pixel groups (128 channels: 4 x 1, 64: 2 x 2, 32: 1 x 4), two workgroups per CU
//     (256 VGPRs per lane, <= 80 KB LDS); an odd tile count (the 64 -> 144 Detect stem: 9 tiles) runs its last tile as a second launch of
//     the TPW = 1 form (1 x 4 waves).
// ------------------------------------------------------------------------------------------------
struct WsGeom { int TH, TW, tiles_x, tiles_y, total_tiles, pitch, nchunks, nwc, nwp; };
#ifdef ZLY_WS_DIAG
__device__ unsigned long long* g_ws_diag = nullptr;              // diagnostic build only (tools/ws_bench.hip): per-wave phase cycle sums
#endif

template <int TPW, int NKS, bool RES, bool ROWT = false, int S = 1>
__global__ __launch_bounds__(256, 2) void conv3x3_ws_kernel(const ConvArgs a, const WsGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lpatch = smem;
    // patch rows are TW + 8 pixels apart although only TW + 2 are used: bank(pixel) repeats every 8 pixels at a pitch of 160 bytes, so a
    // 16-pixel column tile that wraps into the next row (most do at TW = 13) then hits the same banks as 16 consecutive pixels would:
    // conflict-free fragment reads (with PW = TW + 2 three of four reads took 8 LDS cycles instead of 4, and with 18 reads per 36 MFMAs
    // and 8 waves per CU the LDS, not the matrix pipe, set the pace: profiles/r03_ws_kernel_phase_stamps_v2.txt)
    // S = 2 (the down-sampling convs): the patch is (2 TH + 1) x (2 TW + 1) input pixels, a lane's fragments are two patch pixels apart (pixel pitch
    // Cin*2 + 16 B: conflict-free at that stride, tools/lds_pitch.py), and the row pitch keeps PW - TW a multiple of 8 for the same reason as above
    const int PW = S == 1 ? g.TW + 8 : g.TW + 8 * ((g.TW + 8) / 8), PH = S == 1 ? g.TH + 2 : 2 * g.TH + 1;
    const int PWV = S == 1 ? g.TW + 2 : 2 * g.TW + 1;               // patch columns that hold input pixels
    static_assert(S == 1 || !ROWT, "row tiles: stride 1");
    const int nthreads = blockDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kq = lane >> 4;
    const int wc = wave % g.nwc, wp = wave / g.nwc;               // channel group, pixel group of this wave
    const int ntiles_c = (a.cout_pad + 15) >> 4;
    const bf16_t* __restrict__ in = static_cast<const bf16_t*>(a.in) + a.in_co;
    const int upp = a.Cin >> 3;                                 // 16-byte units per patch pixel
    const int NPU = PH * PW * upp;
    const float inv_upp = 1.0f / (float)upp, invPW = 1.0f / (float)PW, invTW = 1.0f / (float)g.TW;
    const int NPB = g.TH * g.TW, nct = ROWT ? g.TH : (NPB + 15) >> 4;          // pixels / 16-pixel column tiles of a tile (ROWT: one tile per row)
    const int tiles_per_img = g.tiles_x * g.tiles_y;

    // ---- this wave's weights: tiles wc * TPW .. + TPW - 1, every k-step, resident in registers for the workgroup's lifetime ----
    bf16x8 w[TPW][NKS];
    {
        const bf16_t* __restrict__ wb = static_cast<const bf16_t*>(a.wgt) + lane * 8;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tile = min(wc * TPW + t, ntiles_c - 1);       // a group's surplus tile (odd tile counts) re-reads the last one; its results are not stored
#pragma unroll
            for (int s = 0; s < NKS; ++s) w[t][s] = *reinterpret_cast<const bf16x8*>(wb + ((size_t)tile * a.nk + s) * 512);
        }
    }
    f32x4 biasr[TPW];
    load_bias<TPW>(a, wc * TPW, kq, biasr);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)((size_t)a.M * a.out_cs * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.out), 0,
                                                                          (unsigned)((size_t)a.M * (a.res ? a.res_cs : a.out_cs) * 2), 0x00020000);
    // this lane's first output channel: TPW = 2 -> a pair of tiles (pair-permuted rows): 8 consecutive channels; TPW = 1 -> 4 channels of tile wc
    // (one tile per wave: the tile may be one of a pair-permuted pair -- 64 channels as 4 waves x 1 tile -- or a plain odd last tile: tile_channel tells)
    const int ch0 = TPW == 2 ? wc * 32 + kq * 8 : tile_channel(wc, kq, (((a.Cout + 15) >> 4) >> 1) << 1);
    constexpr bool has_res = RES;                            // a template parameter: a run-time test would split the block the epilogue shares with the next pair's MFMAs
    // tap offsets of the k-steps inside the patch (wave-uniform): k-step s = tap * nchunks + chunk
    int toff[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        const int tap = s / (NKS / 9), chunk = s - tap * (NKS / 9);
        const int ky = tap / 3, kx = tap - ky * 3;
        toff[s] = (ky * PW + kx) * g.pitch + chunk * 64;
    }
#ifdef ZLY_WS_DIAG
    unsigned long long dsum[4] = {0, 0, 0, 0}, dT0 = 0, dT1 = 0;
#define WSSTAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WSPHASE(k) do { WSSTAMP(dT1); dsum[k] += dT1 - dT0; dT0 = dT1; } while (0)
    const unsigned long long dstart = __builtin_amdgcn_s_memtime();
    WSSTAMP(dT0);
#else
#define WSPHASE(k) do { } while (0)
#endif

    // Patch staging by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write, asynchronous): wave-instruction k fills the 1 KiB
    // of LDS units [64 k, 64 k + 64); lane -> unit u -> (patch pixel, 16-byte piece); the pieces beyond the pixel's channels (the pitch
    // padding), pixels outside the frame (the conv's zero padding) and units beyond the patch get an out-of-range offset: the buffer
    // range check returns zeros.
    const int upitch = g.pitch >> 4;                             // 16-byte units per patch pixel incl. padding
    const int NLU = PH * PW * upitch, ndma = (NLU + 63) >> 6;
    const float inv_upitch = 1.0f / (float)upitch;
    const int patch_bytes = ndma * 1024;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(in), 0, (unsigned)(((size_t)a.M / (a.Ho * a.Wo) * a.H * a.W * a.in_cs - a.in_co) * 2), 0x00020000);
    auto dma_patch = [&](int tl, unsigned char* dst) {
        const int b = tl / tiles_per_img;
        const int r = tl - b * tiles_per_img;
        const int ty = r / g.tiles_x;
        const int y0 = ty * g.TH, x0 = (r - ty * g.tiles_x) * g.TW;
        for (int k = wave; k < ndma; k += nthreads >> 6) {
            const int u = k * 64 + lane;
            const int px = (int)(((float)u + 0.5f) * inv_upitch), part = u - px * upitch;
            const int py = (int)(((float)px + 0.5f) * invPW), pxx = px - py * PW;
            const int gy = y0 * S - 1 + py, gx = x0 * S - 1 + pxx;
            const bool ok = part < upp && pxx < PWV && py < PH && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)((((b * a.H + gy) * a.W + gx) * a.in_cs) * 2 + part * 16) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dst + k * 1024), 16, off, 0, 0, 0);
        }
    };
    (void)NPU; (void)inv_upp; (void)patch_bytes;
    if ((int)blockIdx.x < g.total_tiles) dma_patch(blockIdx.x, lpatch);
    for (int tl = blockIdx.x; tl < g.total_tiles; tl += gridDim.x) {
        const int b = tl / tiles_per_img;
        const int r = tl - b * tiles_per_img;
        const int ty = r / g.tiles_x;
        const int y0 = ty * g.TH, x0 = (r - ty * g.tiles_x) * g.TW;
        unsigned char* cur = lpatch;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of the patch have landed (and its stores of the previous tile are out)
        __syncthreads();                                        // everybody's have
        WSPHASE(0);
        // ---- this wave's column tiles: wp, wp + nwp, ... two at a time (independent accumulators and fragment streams: a lone wave per
        //      workgroup and SIMD has nothing else to cover the ds_read latency with).  The loop is software-pipelined by one pair: the
        //      epilogue of pair k (bias + SiLU + convert: ~150 VALU instructions, the larger half of a pair's time in the stamped build)
        //      stands in the same straight-line block as the 36-72 MFMAs of pair k + 1 -- no branch in between: lanes without a pixel store
        //      to an out-of-range offset -- so the matrix and the vector pipe work at the same time instead of one after the other. ----------------------
        auto pair_setup = [&](int t0, const unsigned char* (&px)[2], int (&ob)[2], int (&rb)[2]) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int t = min(t0 + h2 * g.nwp, nct - 1);
                const int q = ROWT ? (p < g.TW ? t * g.TW + p : NPB) : t * 16 + p;
                const int qc = min(q, NPB - 1);
                const int oy = ROWT ? t : (int)(((float)qc + 0.5f) * invTW), ox = ROWT ? p : qc - oy * g.TW;     // ROWT: lanes beyond the row read patch columns that exist (row pitch TW + 8) and store nothing
                px[h2] = cur + (oy * S * PW + ox * S) * g.pitch + kq * 16;
                const int gy = y0 + oy, gx = x0 + ox;
                const bool ok = (t0 + h2 * g.nwp < nct) && q < NPB && gy < a.Ho && gx < a.Wo;       // a missing second tile: computed on a copy of the last, never stored
                const int m = (b * a.Ho + gy) * a.Wo + gx;
                ob[h2] = ok ? (m * a.out_cs + a.out_co + ch0) * 2 : (int)0x80000000;
                rb[h2] = ok ? (m * a.res_cs + a.res_co + ch0) * 2 : (int)0x80000000;
            }
        };
        auto pair_mma = [&](const unsigned char* const (&px)[2], f32x4 (&acc)[2][TPW]) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int c = 0; c < TPW; ++c) acc[h2][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (ROWT) {
                // One tile = one output ROW (TW + 2 <= 16): lane p reads patch column p of row r + ky ONCE per 32-channel chunk; the fragments
                // of the taps kx = 1, 2 are that of lanes p + 1, p + 2 -- two DPP row shifts instead of two more LDS reads.  With 2 channel tiles
                // per wave the linearised form asks the LDS for 1 KiB per 2 MFMAs (256 B/cycle per CU against the 128 it delivers: the matrix
                // pipe cannot exceed ~50 %); this form needs a third of that, for 13 row tiles instead of 11 column tiles per 13 x 13 tile.
                static_assert(!ROWT || NKS == 18, "row tiles: Cin = 64");
                bf16x8 xr[2][6];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int ck = 0; ck < 2; ++ck) xr[h2][ky * 2 + ck] = *reinterpret_cast<const bf16x8*>(px[h2] + ky * PW * g.pitch + ck * 64);
                auto shl = [](const bf16x8& v, auto ctrl) {
                    const u32x4 u = __builtin_bit_cast(u32x4, v);
                    u32x4 r;
#pragma unroll
                    for (int k = 0; k < 4; ++k) r[k] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u[k], decltype(ctrl)::value, 0xf, 0xf, false);
                    return __builtin_bit_cast(bf16x8, r);
                };
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int ck = 0; ck < 2; ++ck) {
                        bf16x8 f[2][3];
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            f[h2][0] = xr[h2][ky * 2 + ck];
                            f[h2][1] = shl(f[h2][0], std::integral_constant<int, 0x101>{});      // row_shl:1 -- lane p <- lane p + 1
                            f[h2][2] = shl(f[h2][0], std::integral_constant<int, 0x102>{});      // row_shl:2
                        }
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int c = 0; c < TPW; ++c)
#pragma unroll
                                for (int h2 = 0; h2 < 2; ++h2) acc[h2][c] = mma_step(w[c][(ky * 3 + kx) * 2 + ck], f[h2][kx], acc[h2][c]);
                    }
                return;
            }
            // pixel fragments are read WS_DEPTH k-steps ahead of the MFMAs that consume them (statically indexed rings)
            constexpr int WS_DEPTH = 3;
            bf16x8 xf[2][WS_DEPTH + 1];
#pragma unroll
            for (int s = 0; s < WS_DEPTH; ++s) { xf[0][s] = *reinterpret_cast<const bf16x8*>(px[0] + toff[s]); xf[1][s] = *reinterpret_cast<const bf16x8*>(px[1] + toff[s]); }
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                if (s + WS_DEPTH < NKS) {
                    xf[0][(s + WS_DEPTH) % (WS_DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(px[0] + toff[s + WS_DEPTH]);
                    xf[1][(s + WS_DEPTH) % (WS_DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(px[1] + toff[s + WS_DEPTH]);
                }
#pragma unroll
                for (int c = 0; c < TPW; ++c) {
                    acc[0][c] = mma_step(w[c][s], xf[0][s % (WS_DEPTH + 1)], acc[0][c]);
                    acc[1][c] = mma_step(w[c][s], xf[1][s % (WS_DEPTH + 1)], acc[1][c]);
                }
            }
        };
        auto pair_store = [&](const f32x4 (&acc)[2][TPW], const int (&ob)[2], const int (&rb)[2]) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 o[TPW];
#pragma unroll
                for (int c = 0; c < TPW; ++c) {
                    o[c] = acc[h2][c] + biasr[c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[c][r] = silu<bf16_t>(o[c][r]);      // always SiLU here: launch_conv sends a conv without activation elsewhere (a run-time select cost one v_cndmask per value: 16 per pair)
                }
                if (TPW == 2) {
                    f32x4 lo = o[0], hi = o[TPW - 1];
                    if (has_res) {                              // wave-uniform
                        const bf16x8 r = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rres, rb[h2], 0, 0));
                        lo += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
                        hi += f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
                    }
                    const bf16x8 wv = to_bf16x8(lo, hi);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, wv), rout, ob[h2], 0, 0);
                } else {
                    f32x4 x = o[0];
                    if (has_res) {
                        const bf16x4 r = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rres, rb[h2], 0, 0));
                        x += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
                    }
                    const bf16x4 wv = to_bf16x4(x);
                    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, wv), rout, ob[h2], 0, 0);
                }
            }
        };
        if (wp < nct) {
            const unsigned char* pxa[2];
            int oba[2], rba[2];
            f32x4 acca[2][TPW];
            pair_setup(wp, pxa, oba, rba);
            pair_mma(pxa, acca);
            for (int t0 = wp + 2 * g.nwp; t0 < nct; t0 += 2 * g.nwp) {
                const unsigned char* pxb[2];
                int obb[2], rbb[2];
                f32x4 accb[2][TPW];
                pair_setup(t0, pxb, obb, rbb);
                pair_store(acca, oba, rba);                     // epilogue of the previous pair ...
                pair_mma(pxb, accb);                            // ... beside the MFMAs of this one
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    oba[h2] = obb[h2]; rba[h2] = rbb[h2];
#pragma unroll
                    for (int c = 0; c < TPW; ++c) acca[h2][c] = accb[h2][c];
                }
            }
            pair_store(acca, oba, rba);
        }
        WSPHASE(2);
        if (tl + (int)gridDim.x < g.total_tiles) {
            __syncthreads();                                    // everybody is done reading the patch: the next tile's may land (the other resident workgroup computes meanwhile)
            dma_patch(tl + gridDim.x, lpatch);
        }
        WSPHASE(1);
    }
#ifdef ZLY_WS_DIAG
    if (lane == 0 && g_ws_diag) {
        unsigned long long* o = g_ws_diag + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 8;
        for (int k = 0; k < 3; ++k) o[k] = dsum[k];
        o[3] = __builtin_amdgcn_s_memtime() - dstart;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// The 80 -> 80 class-branch convs of Detect (model.22.cv3.L.1; nc = 80), weight-stationary with K PACKED ACROSS TAPS.
//
// Round 3 ran them on conv3x3_lds_kernel with the input stored as 96 channels and a sixth, all-zero output tile (27 k-steps x 6 tiles): 31 % of its
// MFMAs and of its fragment traffic were on zeros (46.6 us at P3 for 8.8 us of attainable work).  conv3x3_ws_kernel cannot take them: two
// tiles per wave x 27 k-steps is 216 weight registers, and one tile per wave leaves 5 tiles over 4 waves.  Here:
//   * K runs over (tap, channel) without padding: k = tap * 80 + c, 720 = 22.5 k-steps of 32 -> NKS = 23.  Channels come in groups of 8 (one 16-byte
//     fragment piece), 10 per tap, so a k-step's four lane groups sit in up to two taps: lane group kq of k-step s reads group G = 4 s + kq =
//     (tap G / 10, piece G % 10) -- a PER-LANE patch offset, precomputed once (two 16-bit offsets per register);
//   * a workgroup is FIVE waves, one per 16-channel output tile (plain tile rows: 8-byte NHWC stores), all walking the same pixel tile past their 92
//     weight registers; two workgroups per CU (<= 168 VGPRs: three waves per SIMD);
//   * everything else is conv3x3_ws_kernel: patch by LDS-DMA with the buffer range check as zero padding (pitch 80 * 2 B: wsk_plan), persistent over 13 x 13-ish
//     tiles, column tiles in pairs, epilogue of a pair beside the next pair's MFMAs.
// The input view is the first 80 of the 96 stored channels; the weights are tiled with cin_store = 80 (weights.cpp: k = tap * 80 + c, zero beyond 720).
// ------------------------------------------------------------------------------------------------
template <int NKS>
__global__ __launch_bounds__(320, 3) void conv3x3_wsk_kernel(const ConvArgs a, const WsGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lpatch = smem;
    const int PW = g.TW + 8, PH = g.TH + 2, PWV = g.TW + 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int p = lane & 15, kq = lane >> 4;
    const bf16_t* __restrict__ in = static_cast<const bf16_t*>(a.in) + a.in_co;
    const int upp = a.Cin >> 3;                                 // 16-byte pieces per pixel (10)
    const float invPW = 1.0f / (float)PW, invTW = 1.0f / (float)g.TW;
    const int NPB = g.TH * g.TW, nct = (NPB + 15) >> 4;
    const int tiles_per_img = g.tiles_x * g.tiles_y;
    const int ntiles_c = (a.Cout + 15) >> 4;
    const int tile = min(wave, ntiles_c - 1);

    bf16x8 w[NKS];
    {
        const bf16_t* __restrict__ wb = static_cast<const bf16_t*>(a.wgt) + lane * 8;
#pragma unroll
        for (int s = 0; s < NKS; ++s) w[s] = *reinterpret_cast<const bf16x8*>(wb + ((size_t)tile * a.nk + s) * 512);
    }
    const int ch0 = tile * 16 + kq * 4;
    const f32x4 biasr = ch0 < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + ch0) : f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)((size_t)a.M * a.out_cs * 2), 0x00020000);
    // this lane group's patch offset per k-step, two per register: group G = 4 s + kq -> tap G / 10 (taps beyond the ninth: zero weights, offset 0)
    unsigned toff2[(NKS + 1) / 2];
#pragma unroll
    for (int s2 = 0; s2 < (NKS + 1) / 2; ++s2) {
        unsigned v = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s = s2 * 2 + h;
            const int G = 4 * s + kq;
            const int tap = (G * 205) >> 11;                    // G / 10 for G < 1024
            const int c8 = G - tap * 10;
            const int ky = (tap * 11) >> 5, kx = tap - ky * 3;  // tap / 3 for tap < 32
            const unsigned o = (s < NKS && tap < 9) ? (unsigned)((ky * PW + kx) * g.pitch + c8 * 16) : 0u;
            v |= o << (16 * h);
        }
        toff2[s2] = v;
    }
    auto toff = [&](int s) -> unsigned { return (s & 1) ? toff2[s >> 1] >> 16 : toff2[s >> 1] & 0xffffu; };

    const int upitch = g.pitch >> 4;
    const int NLU = PH * PW * upitch, ndma = (NLU + 63) >> 6;
    const float inv_upitch = 1.0f / (float)upitch;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(in), 0, (unsigned)(((size_t)a.M / (a.Ho * a.Wo) * a.H * a.W * a.in_cs - a.in_co) * 2), 0x00020000);
    auto dma_patch = [&](int tl, unsigned char* dst) {
        const int b = tl / tiles_per_img;
        const int r = tl - b * tiles_per_img;
        const int ty = r / g.tiles_x;
        const int y0 = ty * g.TH, x0 = (r - ty * g.tiles_x) * g.TW;
        for (int k = wave; k < ndma; k += nwaves) {
            const int u = k * 64 + lane;
            const int px = (int)(((float)u + 0.5f) * inv_upitch), part = u - px * upitch;
            const int py = (int)(((float)px + 0.5f) * invPW), pxx = px - py * PW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + pxx;
            const bool ok = part < upp && pxx < PWV && py < PH && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)((((b * a.H + gy) * a.W + gx) * a.in_cs) * 2 + part * 16) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dst + k * 1024), 16, off, 0, 0, 0);
        }
    };
    if ((int)blockIdx.x < g.total_tiles) dma_patch(blockIdx.x, lpatch);
    for (int tl = blockIdx.x; tl < g.total_tiles; tl += gridDim.x) {
        const int b = tl / tiles_per_img;
        const int r = tl - b * tiles_per_img;
        const int ty = r / g.tiles_x;
        const int y0 = ty * g.TH, x0 = (r - ty * g.tiles_x) * g.TW;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of the patch have landed (and its stores of the previous tile are out)
        __syncthreads();
        auto pair_setup = [&](int t0, const unsigned char* (&px)[2], int (&ob)[2]) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int t = min(t0 + h2, nct - 1);
                const int q = t * 16 + p;
                const int qc = min(q, NPB - 1);
                const int oy = (int)(((float)qc + 0.5f) * invTW), ox = qc - oy * g.TW;
                px[h2] = lpatch + (oy * PW + ox) * g.pitch;
                const int gy = y0 + oy, gx = x0 + ox;
                const bool ok = (t0 + h2 < nct) && q < NPB && gy < a.Ho && gx < a.Wo && ch0 < a.Cout && wave < ntiles_c;
                const int m = (b * a.Ho + gy) * a.Wo + gx;
                ob[h2] = ok ? (m * a.out_cs + a.out_co + ch0) * 2 : (int)0x80000000;
            }
        };
        auto pair_mma = [&](const unsigned char* const (&px)[2], f32x4 (&acc)[2]) {
            acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr int DEPTH = 2;
            bf16x8 xf[2][DEPTH + 1];
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) { xf[0][s] = *reinterpret_cast<const bf16x8*>(px[0] + toff(s)); xf[1][s] = *reinterpret_cast<const bf16x8*>(px[1] + toff(s)); }
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                if (s + DEPTH < NKS) {
                    xf[0][(s + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(px[0] + toff(s + DEPTH));
                    xf[1][(s + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(px[1] + toff(s + DEPTH));
                }
                acc[0] = mma_step(w[s], xf[0][s % (DEPTH + 1)], acc[0]);
                acc[1] = mma_step(w[s], xf[1][s % (DEPTH + 1)], acc[1]);
            }
        };
        auto pair_store = [&](const f32x4 (&acc)[2], const int (&ob)[2]) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 o = acc[h2] + biasr;
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2) o[r2] = silu<bf16_t>(o[r2]);       // launch_conv_wsk refuses a conv without activation
                const bf16x4 wv = to_bf16x4(o);
                typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, wv), rout, ob[h2], 0, 0);
            }
        };
        {
            const unsigned char* pxa[2];
            int oba[2];
            f32x4 acca[2];
            pair_setup(0, pxa, oba);
            pair_mma(pxa, acca);
            for (int t0 = 2; t0 < nct; t0 += 2) {
                const unsigned char* pxb[2];
                int obb[2];
                f32x4 accb[2];
                pair_setup(t0, pxb, obb);
                pair_store(acca, oba);                          // epilogue of the previous pair beside the MFMAs of this one
                pair_mma(pxb, accb);
                oba[0] = obb[0]; oba[1] = obb[1]; acca[0] = accb[0]; acca[1] = accb[1];
            }
            pair_store(acca, oba);
        }
        if (tl + (int)gridDim.x < g.total_tiles) {
            __syncthreads();                                    // everybody is done reading the patch: the next tile's may land
            dma_patch(tl + gridDim.x, lpatch);
        }
    }
}

// tile plan of the K-packed kernel: as ws_plan, for a pixel pitch of exactly cin * 2 = 160 bytes (ten 16-byte pieces, no padding).  That is the pitch
// conv3x3_ws_kernel uses for 64 channels (128 + 32), and it is conflict-free here for the same reason: a 16-lane group of ds_read_b128 holds 8 pixels of
// one lane group (pieces 10 p: all even slots of the 256-byte bank row) and 8 of its neighbour, whose offset is ONE piece further -- the next piece of the
// same tap, or, where a k-step straddles two taps, piece 0 of the next pixel, which in an unpadded row is again exactly one piece on.  (First version:
// 160 + 32 = 192 bytes = 12 pieces: 12 p mod 16 has period 4, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.51, no faster than the LDS-tiled kernel.)
static bool wsk_plan(int H, int W, int cin, int n, WsGeom* g)
{
    const int pitch = cin * 2;
    long best = -1;
    for (int th = 4; th <= 32; ++th)
        for (int tw = 8; tw <= 32; ++tw) {
            const long ph = th + 2, pw = tw + 8;
            const long lds = (ph * pw * pitch + 1023) / 1024 * 1024;
            if (lds > 64 * 1024) continue;
            const int tx = (W + tw - 1) / tw, ty = (H + th - 1) / th;
            const long tiles = (long)tx * ty * n;
            const long rounds = (tiles + 2 * num_cus() - 1) / (2 * num_cus());
            const long key = rounds * ((th * tw + 15) / 16 * 16 + 64) * 4096 + ph * pw;
            if (best < 0 || key < best) { best = key; g->TH = th; g->TW = tw; g->tiles_x = tx; g->tiles_y = ty; }
        }
    if (best < 0) return false;
    g->pitch = pitch; g->nchunks = 0;
    return true;
}

// may this conv take it?  (3x3 stride 1, 80 real input channels, 80 output channels, enough pixels for the persistent grid, tiles that cover the map well)
bool conv_wsk_ok(int cin, int cout, int n, int Ho, int Wo)
{
    if (cin != 80 || cout != 80) return false;
    WsGeom g{};
    if (!wsk_plan(Ho, Wo, cin, n, &g)) return false;
    const double util = (double)Ho * Wo / ((double)g.tiles_x * g.tiles_y * g.TH * g.TW);
    const char* mt = getenv("ZLY_WS_MIN_TILES");
    return util >= 0.7 && (long)n * Ho * Wo >= (mt ? atol(mt) : 64) * 169L;
}

hipError_t launch_conv_wsk(const ConvArgs& a, hipStream_t s)
{
    WsGeom g{};
    const int n = a.M / (a.Ho * a.Wo);
    if (a.Cin != 80 || a.Cout != 80 || a.nk != 23 || a.stride != 1 || a.pad != 1 || a.in2 || a.res || a.out_f32 || !a.act || a.in_cs % 8 || a.in_co % 8 || a.out_cs % 4 || a.out_co % 4 ||
        !wsk_plan(a.Ho, a.Wo, a.Cin, n, &g)) return hipErrorInvalidValue;
    if ((size_t)a.M * (size_t)std::max(a.in_cs, a.out_cs) * 2 >= ((size_t)1 << 31)) return hipErrorInvalidValue;        // 32-bit buffer offsets (the caller falls back)
    g.total_tiles = g.tiles_x * g.tiles_y * n;
    g.nwc = 5; g.nwp = 1;
    const size_t lds = ((size_t)(g.TH + 2) * (g.TW + 8) * g.pitch + 1023) / 1024 * 1024;
    const int gx = g.total_tiles < 2 * num_cus() ? g.total_tiles : 2 * num_cus();
    hipLaunchKernelGGL(conv3x3_wsk_kernel<23>, dim3(gx), dim3(320), lds, s, a, g);
    return hipGetLastError();
}

typedef void (*conv_ws_fn)(const ConvArgs, const WsGeom);
static conv_ws_fn pick_ws(int cin, int tpw, bool res = false, bool rowt = false, int stride = 1)
{
    if (stride == 2) {          // the down-sampling convs with 32 / 64 input channels and an even number of output tiles; no residual
        if (res || rowt) return nullptr;
        if (tpw == 1) return cin == 64 ? conv3x3_ws_kernel<1, 18, false, false, 2> : nullptr;      // 64 -> 64 on small pixel tiles: 4 waves x one tile
        if (tpw != 2) return nullptr;
        return cin == 64 ? conv3x3_ws_kernel<2, 18, false, false, 2> : cin == 32 ? conv3x3_ws_kernel<2, 9, false, false, 2> : nullptr;
    }
    if (cin != 64) return nullptr;
    if (rowt && tpw == 2) return res ? conv3x3_ws_kernel<2, 18, true, true> : conv3x3_ws_kernel<2, 18, false, true>;
    if (res) return tpw == 2 ? conv3x3_ws_kernel<2, 18, true> : conv3x3_ws_kernel<1, 18, true>;
    return tpw == 2 ? conv3x3_ws_kernel<2, 18, false> : conv3x3_ws_kernel<1, 18, false>;
}
static constexpr int WS_LDS_MAX = 64 * 1024;        // two resident workgroups per CU with room to spare (two of exactly 80 KB did not both become resident)

// tile shape: among the shapes whose patch lets two workgroups be resident per CU, the one with the least work on the busiest workgroup --
// rounds of tiles over the 2 x 256 resident workgroups x (pixels of a tile + a fixed per-tile cost); 768 tiles of 9 x 26 lose to 1024 of 13 x 13
static bool ws_plan(int H, int W, int cin, int n, WsGeom* g, int stride = 1)
{
    if (!pick_ws(cin, 2, false, false, stride)) return false;
    const int pitch = cin * 2 + (stride == 2 ? 16 : 32);
    long best = -1;
    for (int th = (stride == 2 ? 2 : 4); th <= 32; ++th)
        for (int tw = 8; tw <= 32; ++tw) {
            const long ph = stride == 2 ? 2 * th + 1 : th + 2, pw = stride == 2 ? tw + 8 * ((tw + 8) / 8) : tw + 8;
            const long lds = (ph * pw * pitch + 1023) / 1024 * 1024;
            if (lds > WS_LDS_MAX) continue;
            const int tx = (W + tw - 1) / tw, ty = (H + th - 1) / th;
            const long tiles = (long)tx * ty * n;
            const long rounds = (tiles + 2 * num_cus() - 1) / (2 * num_cus());
            const long key = rounds * ((th * tw + 15) / 16 * 16 + 64) * 4096 + ph * pw;
            if (best < 0 || key < best) { best = key; g->TH = th; g->TW = tw; g->tiles_x = tx; g->tiles_y = ty; }
        }
    if (best < 0) return false;
    g->pitch = pitch; g->nchunks = cin / 32;
    return true;
}

hipError_t ws_init()
{
    {
        hipError_t r = hipFuncSetAttribute((const void*)conv3x3_wsk_kernel<23>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_MAX);
        if (r != hipSuccess) return r;
    }
    for (int tpw = 1; tpw <= 2; ++tpw)
        for (int res = 0; res <= 1; ++res)
            for (int rowt = 0; rowt <= 1; ++rowt) {
                hipError_t r = hipFuncSetAttribute((const void*)pick_ws(64, tpw, res != 0, rowt != 0), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_MAX);
                if (r != hipSuccess) return r;
            }
    for (int cin = 32; cin <= 64; cin += 32) {
        hipError_t r = hipFuncSetAttribute((const void*)pick_ws(cin, 2, false, false, 2), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_MAX);
        if (r != hipSuccess) return r;
    }
    {
        hipError_t r = hipFuncSetAttribute((const void*)pick_ws(64, 1, false, false, 2), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_MAX);
        if (r != hipSuccess) return r;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolution, WEIGHT-STATIONARY PER WAVE (bf16, single source, SiLU, Cin = 32 * NK): the K-heavy pointwise layers (C2f cv1 / cv2 and
// SPPF with 128 .. 1024 input channels).  In the direct kernel a wave walks K with fragments two k-steps in flight: its launch lasts as
// long as ONE wave's chain of NK / 2 dependent L2 round trips (13 - 18 us for 1.4 - 3.9 GFLOP at batch 64; 205 - 340 TFLOP/s on the
// YOLOv8-s layers whatever their size).  Here nothing is dependent: a wave loads the weight fragments of its TPW channel tiles for ALL of
// K into registers once (TPW x NK x 4 VGPRs), the workgroup's pixel tile (NPX consecutive NHWC pixels, all of K) lands in LDS by LDS-DMA in
// one go, and each wave streams the 16-pixel column tiles past its weights exactly as conv3x3_ws_kernel does (pairs of column tiles,
// epilogue of one pair beside the MFMAs of the next, no barrier inside a tile).  Workgroup = 4 waves = NWC channel groups x NWP pixel
// groups; blockIdx.y = block of NWC x TPW channel tiles; persistent over pixel tiles, two workgroups per CU.
// ------------------------------------------------------------------------------------------------
struct Ws1Geom { int npx, total_tiles, pitch, nwc, nwp; };

template <int TPW, int NK, bool DUAL = false>
__global__ __launch_bounds__(256, 2) void conv1x1_ws_kernel(const ConvArgs a, const Ws1Geom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kq = lane >> 4;
    const int wc = wave % g.nwc, wp = wave / g.nwc;               // channel group, pixel group of this wave
    const int tile0 = ((int)blockIdx.y * g.nwc + wc) * TPW;       // this wave's first 16-channel tile
    const bf16_t* __restrict__ in = static_cast<const bf16_t*>(a.in) + a.in_co;

    bf16x8 w[TPW][NK];
    {
        const bf16_t* __restrict__ wb = static_cast<const bf16_t*>(a.wgt) + lane * 8;
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int s = 0; s < NK; ++s) w[t][s] = *reinterpret_cast<const bf16x8*>(wb + ((size_t)(tile0 + t) * a.nk + s) * 512);
    }
    f32x4 biasr[TPW];
    load_bias<TPW>(a, tile0, kq, biasr);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)((size_t)a.M * a.out_cs * 2), 0x00020000);
    // DUAL (fused nearest-2x Upsample + Concat, see ConvArgs): `in` is the half-size tensor, `in2` the full-size one
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(in), 0, (unsigned)(((size_t)(DUAL ? a.M >> 2 : a.M) * a.in_cs - a.in_co) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rin2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(DUAL ? a.in2 : a.in), 0, (unsigned)((size_t)a.M * (DUAL ? a.in2_cs : a.in_cs) * 2), 0x00020000);
    // this lane's first output channel: TPW = 2 -> a pair of tiles (pair-permuted rows): 8 consecutive channels; TPW = 1 -> 4 channels
    const int ch0 = TPW == 2 ? (tile0 >> 1) * 32 + kq * 8 : tile_channel(tile0, kq, (((a.Cout + 15) >> 4) >> 1) << 1);

    // pixel tile -> LDS by LDS-DMA: wave-instruction k fills the LDS units [64 k, 64 k + 64); lane -> unit -> (pixel, 16-byte piece); the
    // pitch padding, pixels beyond M and units beyond the tile get an out-of-range offset (zeros)
    const int upp = a.Cin >> 3, upitch = g.pitch >> 4;
    const int ndma = (g.npx * upitch + 63) >> 6;
    const float inv_upitch = 1.0f / (float)upitch;
    const int hw = a.H * a.W, spa = a.split_c >> 3;              // DUAL: pixels of a frame, 16-byte pieces that come from `in`
    const float inv_hw = 1.0f / (float)hw, invW = 1.0f / (float)a.W;
    auto dma_patch = [&](int tl) {
        const int m0 = tl * g.npx;
        for (int k = wave; k < ndma; k += 4) {
            const int u = k * 64 + lane;
            const int q = (int)(((float)u + 0.5f) * inv_upitch), part = u - q * upitch;
            const bool ok = part < upp && q < g.npx && m0 + q < a.M;
            auto* dst = (__attribute__((address_space(3))) void*)(smem + k * 1024);
            if constexpr (!DUAL) {
                const unsigned off = ok ? (unsigned)((m0 + q) * a.in_cs * 2 + part * 16) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, dst, 16, off, 0, 0, 0);
            } else {
                // two DMA instructions under complementary EXEC masks (a masked-off lane writes nothing): the pieces of the up-sampled
                // tensor (and the zero pieces), then those of the full-size one
                const int m = min(m0 + q, a.M - 1);
                const int b = (int)(((float)m + 0.5f) * inv_hw), r = m - b * hw;
                const int y = (int)(((float)r + 0.5f) * invW), x = r - y * a.W;
                const int ma = (b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1);
                if (!ok || part < spa) {
                    const unsigned off = ok ? (unsigned)(ma * a.in_cs * 2 + part * 16) : 0x80000000u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, dst, 16, off, 0, 0, 0);
                } else {
                    const unsigned off = (unsigned)((m * a.in2_cs + a.in2_co) * 2 + (part - spa) * 16);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin2, dst, 16, off, 0, 0, 0);
                }
            }
        }
    };
    (void)rin2; (void)inv_hw; (void)invW; (void)spa;
    if ((int)blockIdx.x < g.total_tiles) dma_patch(blockIdx.x);
    for (int tl = blockIdx.x; tl < g.total_tiles; tl += gridDim.x) {
        const int m0 = tl * g.npx;
        const int nct = (min(g.npx, a.M - m0) + 15) >> 4;         // column tiles with at least one pixel
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of the tile have landed (and its stores of the previous tile are out)
        __syncthreads();
        auto pair_setup = [&](int t0, const unsigned char* (&px)[2], int (&ob)[2]) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int t = min(t0 + h2 * g.nwp, nct - 1);
                const int q = t * 16 + p, m = m0 + q;
                px[h2] = smem + q * g.pitch + kq * 16;
                const bool ok = (t0 + h2 * g.nwp < nct) && m < a.M;     // a missing second tile: computed on a copy of the last, never stored
                ob[h2] = ok ? (m * a.out_cs + a.out_co + ch0) * 2 : (int)0x80000000;
            }
        };
        auto pair_mma = [&](const unsigned char* const (&px)[2], f32x4 (&acc)[2][TPW]) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int c = 0; c < TPW; ++c) acc[h2][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr int DEPTH = 3;                            // pixel fragments are read DEPTH k-steps ahead of their MFMAs
            bf16x8 xf[2][DEPTH + 1];
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) { xf[0][s] = *reinterpret_cast<const bf16x8*>(px[0] + s * 64); xf[1][s] = *reinterpret_cast<const bf16x8*>(px[1] + s * 64); }
#pragma unroll
            for (int s = 0; s < NK; ++s) {
                if (s + DEPTH < NK) {
                    xf[0][(s + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(px[0] + (s + DEPTH) * 64);
                    xf[1][(s + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(px[1] + (s + DEPTH) * 64);
                }
#pragma unroll
                for (int c = 0; c < TPW; ++c) {
                    acc[0][c] = mma_step(w[c][s], xf[0][s % (DEPTH + 1)], acc[0][c]);
                    acc[1][c] = mma_step(w[c][s], xf[1][s % (DEPTH + 1)], acc[1][c]);
                }
            }
        };
        auto pair_store = [&](const f32x4 (&acc)[2][TPW], const int (&ob)[2]) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 o[TPW];
#pragma unroll
                for (int c = 0; c < TPW; ++c) {
                    o[c] = acc[h2][c] + biasr[c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[c][r] = silu<bf16_t>(o[c][r]);
                }
                if (TPW == 2) {
                    const bf16x8 wv = to_bf16x8(o[0], o[TPW - 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, wv), rout, ob[h2], 0, 0);
                } else {
                    const bf16x4 wv = to_bf16x4(o[0]);
                    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, wv), rout, ob[h2], 0, 0);
                }
            }
        };
        if (wp < nct) {
            const unsigned char* pxa[2];
            int oba[2];
            f32x4 acca[2][TPW];
            pair_setup(wp, pxa, oba);
            pair_mma(pxa, acca);
            for (int t0 = wp + 2 * g.nwp; t0 < nct; t0 += 2 * g.nwp) {
                const unsigned char* pxb[2];
                int obb[2];
                f32x4 accb[2][TPW];
                pair_setup(t0, pxb, obb);
                pair_store(acca, oba);                          // epilogue of the previous pair beside the MFMAs of this one
                pair_mma(pxb, accb);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    oba[h2] = obb[h2];
#pragma unroll
                    for (int c = 0; c < TPW; ++c) acca[h2][c] = accb[h2][c];
                }
            }
            pair_store(acca, oba);
        }
        if (tl + (int)gridDim.x < g.total_tiles) {
            __syncthreads();                                    // everybody is done reading the tile: the next one may land
            dma_patch(tl + gridDim.x);
        }
    }
}

typedef void (*conv_ws1_fn)(const ConvArgs, const Ws1Geom);
static conv_ws1_fn pick_ws1(int tpw, int nk, bool dual = false)
{
#define ZLY_WS1_CASE(T_, N_) if (tpw == T_ && nk == N_) return conv1x1_ws_kernel<T_, N_>
    if (dual) {                 // the Upsample + Concat inputs of the neck: 192 / 384 channels (YOLOv8n), 384 (YOLOv8-s)
        if (tpw == 2 && nk == 6) return conv1x1_ws_kernel<2, 6, true>;
        if (tpw == 2 && nk == 12) return conv1x1_ws_kernel<2, 12, true>;
        return nullptr;         // 768 channels (one channel tile per wave, four channel blocks that each gather the input): the direct kernel is faster (58 vs 70 us)
    }
    ZLY_WS1_CASE(2, 4); ZLY_WS1_CASE(2, 6); ZLY_WS1_CASE(2, 8); ZLY_WS1_CASE(2, 12); ZLY_WS1_CASE(2, 16);
    ZLY_WS1_CASE(1, 24); ZLY_WS1_CASE(1, 32);
#undef ZLY_WS1_CASE
    return nullptr;
}
static int ws1_tpw(int nk) { return nk <= 16 ? 2 : 1; }

// pixel tile: as many pixels as 64 KB of LDS hold (two workgroups per CU), fewer while the launch has less than two workgroups per CU
static bool ws1_plan(int cin, int cout_pad, int M, Ws1Geom* g, int* ny, bool dual = false)
{
    const int nk = cin / 32, tpw = ws1_tpw(nk);
    if (cin % 32 || !pick_ws1(tpw, nk, dual)) return false;
    const int ntiles = cout_pad / 16;
    if (cout_pad % 32) return false;
    const int per = ntiles / tpw;                                 // channel groups (waves' worth) in all
    g->nwc = per % 4 == 0 ? 4 : per % 2 == 0 ? 2 : 1;
    g->nwp = 4 / g->nwc;
    *ny = per / g->nwc;
    g->pitch = cin * 2 + 32;                                      // conflict-free ds_read_b128 for every Cin % 32 == 0 (tools/lds_pitch.py)
    // pixels per tile: the least work on the busiest of the 2 x 256 resident workgroups -- rounds of tiles x (pixels of a tile + a fixed
    // per-tile cost: weights, DMA latency, barriers ~ 128 pixels' worth); at most what 64 KB of LDS hold
    int npx_max = WS_LDS_MAX / g->pitch / 16 * 16;                 // (smaller caps, for more co-resident workgroups of other chains, measured: 32 / 48 KB no gain, 16 - 24 KB -1.5 %)
    if (npx_max > 256) npx_max = 256;
    if (npx_max < 16) return false;
    long best = -1;
    for (int npx = 16 * g->nwp; npx <= npx_max; npx += 16) {
        const long tiles = (long)((M + npx - 1) / npx) * *ny;
        const long rounds = (tiles + 2L * num_cus() - 1) / (2L * num_cus());
        const long key = rounds * (npx + 128);
        if (best < 0 || key < best) { best = key; g->npx = npx; }
    }
    if (best < 0) return false;
    g->total_tiles = (M + g->npx - 1) / g->npx;
    return true;
}

hipError_t ws1_init()
{
    static const int shapes[9][3] = {{2, 4, 0}, {2, 6, 0}, {2, 8, 0}, {2, 12, 0}, {2, 16, 0}, {1, 24, 0}, {1, 32, 0}, {2, 6, 1}, {2, 12, 1}};
    for (int i = 0; i < 9; ++i) {
        hipError_t r = hipFuncSetAttribute((const void*)pick_ws1(shapes[i][0], shapes[i][1], shapes[i][2] != 0), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (r != hipSuccess) return r;
    }
    return hipSuccess;
}

static int g_num_cus = 256;
int num_cus() { return g_num_cus; }
void set_num_cus(int n) { g_num_cus = n < 1 ? 1 : (n > 256 ? 256 : n); }

// dynamic LDS above 64 KiB needs an opt-in per kernel; done once, outside any stream capture
hipError_t conv_init()
{
    { hipError_t r = ws_init(); if (r != hipSuccess) return r; }
    { hipError_t r = ws1_init(); if (r != hipSuccess) return r; }
    static const int pts1[3] = {1, 2, 4}, pts2[2] = {1, 2};
    for (int ct = 2; ct <= 5; ++ct) {
        for (int i = 0; i < 3; ++i) {
            hipError_t r = hipFuncSetAttribute((const void*)pick_lds(1, pts1[i], ct), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(1, pts1[i], ct, 2));
            if (r != hipSuccess) return r;
        }
        for (int i = 0; i < 2; ++i) {
            hipError_t r = hipFuncSetAttribute((const void*)pick_lds(2, pts2[i], ct), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(2, pts2[i], ct));
            if (r != hipSuccess) return r;
        }
    }
    return hipSuccess;
}

typedef void (*conv_fn)(const ConvArgs);

template <typename T, int MODE, int PT, int KSPLIT>
static conv_fn pick_ct(int ct) {
    switch (ct) {
        case 1: return conv_igemm_kernel<T, MODE, 1, PT, KSPLIT>;
        case 2: return conv_igemm_kernel<T, MODE, 2, PT, KSPLIT>;
        case 3: return conv_igemm_kernel<T, MODE, 3, PT, KSPLIT>;
        case 4: return conv_igemm_kernel<T, MODE, 4, PT, KSPLIT>;
        case 5: return conv_igemm_kernel<T, MODE, 5, PT, KSPLIT>;
    }
    return nullptr;
}
template <typename T, int PT, int KSPLIT>
static conv_fn pick_mode(int mode, int ct) {
    switch (mode) {
        case 0: return pick_ct<T, 0, PT, KSPLIT>(ct);
        case 1: return pick_ct<T, 1, PT, KSPLIT>(ct);
        case 2: return pick_ct<T, 2, PT, KSPLIT>(ct);
    }
    return nullptr;
}

void conv_pick_direct(int dtype, int ks, int cin, int cout_pad, int M, ConvLaunch* cfg);

int conv_kstep(int dtype) { return dtype == ZLY_DTYPE_BF16 ? Frag<bf16_t>::KSTEP : Frag<float>::KSTEP; }

// Tile shape per launch.  The chip has 256 CUs; a launch wants >= ~2 workgroups per CU.
//   large M (batch 64, shallow layers): PT = 4 / 2 pixel tiles per wave amortise the weight fragments;
//   small M (batch 1, deep layers): PT = 1, fewer channel tiles per wave and 4-way split-K, so that
//   even the 13x13 layers put a few hundred workgroups on the chip.
// LDS-tiled 3x3 kernel when the launch is big enough to fill the chip with TH x 16 tiles
static bool pick_lds_config(int stride, int cin, int cout_pad, int n, int Ho, int Wo, ConvLaunch* cfg)
{
    if (cin % 32 != 0) return false;
    const int ntiles = cout_pad / 16;
    int ct = 0;
    static const int pref[4] = {4, 5, 3, 2};
    for (int i = 0; i < 4; ++i)
        if (ntiles % pref[i] == 0) { ct = pref[i]; break; }
    if (!ct) return false;
    // resident weights: Cin = 32 (one chunk: every item of the workgroup uses the same tiles, so they are staged once).  For
    // Cin = 64 with CT = 3 (the 64 -> 144 Detect stems, 54 KB for both chunks) residency won 12 % while weights went through
    // registers; since they go by LDS-DMA the per-item copy is cheap and three resident workgroups beat two with resident
    // weights (ZLY_WRES_MAXCHUNKS=2 restores it).  Halving CT to make 64 -> 64 resident was always slower (37 vs 32 us).
    cfg->wres = 0;
    const char* nr = getenv("ZLY_NO_WRES");
    const char* wm = getenv("ZLY_WRES_MAXCHUNKS");                   // tuning aid
    if (stride == 1 && cin <= 32 * (wm ? atoi(wm) : 1) && !nr) {
        static const int rpref[4] = {4, 3, 2, 5};
        for (int i = 0; i < 4; ++i) {
            const int c = rpref[i];
            if (ntiles % c == 0 && c >= ct && (size_t)(cin / 32) * 9 * c * 1024 + LdsGeom<1, 2>::PATCH_BYTES <= 80 * 1024) { ct = c; cfg->wres = 1; break; }
        }
    }
    const int ytiles = ntiles / ct;
    const int tx = (Wo + 15) / 16;
    // stride 2 (round 3): 8-row tiles where they still leave enough work items for the persistent grid (model.5 25.1 -> 21.6 us, model.7 23.6 -> 21.1;
    // model.19's 256 items stay on 4-row tiles)
    int pt = stride == 1 ? 2 : (((long)n * ((Wo + 15) / 16) * ((Ho + 7) / 8) * (ntiles / ct) >= 384 && !getenv("ZLY_LDS_S2_PT1")) ? 2 : 1);
    if (stride == 1 && ct == 2 && (long)n * tx * ((Ho + 15) / 16) * ytiles >= 2048) pt = 4;   // plenty of tiles, few channels: bigger tiles (VGPR budget)
    if (stride == 1 && Ho <= 14) pt = 1;                                            // 13-row maps: 4 x 4 rows
    if (ct == 5 && (pt > 1 || stride == 2)) { if (stride == 2) return false; pt = 1; }  // those variants do not fit 256 registers (2 waves per SIMD) without spilling
    const long tiles = (long)n * tx * ((Ho + 4 * pt - 1) / (4 * pt));
    const char* lm = getenv("ZLY_LDS_MIN_TILES");                                   // tuning / tests: force the LDS kernel onto small launches
    if (tiles * ytiles < (lm ? atol(lm) : 384)) return false;                       // too small: direct kernel
    cfg->lds = 1; cfg->ct = ct; cfg->pt = pt; cfg->ksplit = 1; cfg->fastk = 1;
    return true;
}

// streaming 1x1 kernel: single-source pointwise convs with enough pixels to keep persistent waves busy
static bool pick_stream_config(int cin, int cout_pad, int M, ConvLaunch* cfg)
{
    const int nk = (cin + 31) / 32;
    const int ntiles = cout_pad / 16;
    int ct = 0, pt = 0;
    if (nk <= 2 && ntiles == 2) { ct = 2; pt = 4; }
    else if (nk <= 4 && ntiles % 4 == 0) { ct = 4; pt = 2; }
    // 192 / 256 input channels (round 3): 64 output channels per wave (CT = 4) -- with CT = 2 the four channel blocks of a 128-channel layer
    // each read the whole input: model.12.cv2 17.7 -> 15.1 us, model.18.cv1 / cv2 16.7 -> 14.5.  256 input channels go to the direct kernel,
    // which is faster still there (model.6.cv2 21.1 -> 16.9 us, model.8.cv1 15.3 -> 13.3).  ZLY_STREAM_CT2 restores the old shapes (tests).
    else if (nk == 6 && ntiles % 4 == 0 && !getenv("ZLY_STREAM_CT2")) { ct = 4; pt = 1; }
    else if (nk == 8 && !getenv("ZLY_STREAM_CT2")) return false;
    else if ((nk == 6 || nk == 8) && ntiles % 2 == 0) { ct = 2; pt = 1; }
    if (!ct || !pick_stream(ct, pt, nk)) return false;
    if (const char* mx = getenv("ZLY_STREAM_MAX_NK")) { if (nk > atoi(mx)) return false; }          // tuning aid
    const char* mg = getenv("ZLY_STREAM_MIN_GROUPS");                     // tuning / tests: force the streaming kernel onto small launches
    if ((long)M / (16 * pt) * (ntiles / ct) < (mg ? atol(mg) : 4096)) return false;   // too few pixel groups to keep persistent waves busy
    cfg->stream = 1; cfg->ct = ct; cfg->pt = pt; cfg->ksplit = 1; cfg->fastk = 0; cfg->lds = 0;
    return true;
}

// weight-stationary 3x3 kernel: stride-1 layers with Cin = 64 and 32 / 64 / 128 (+ 16) output channels on maps its tiles cover well, with
// enough tiles to fill the chip
static bool pick_ws_config(int stride, int cin, int cout_pad, int n, int Ho, int Wo, ConvLaunch* cfg)
{
    const bool off = getenv("ZLY_NO_WS") != nullptr;               // tuning / tests (read per picked shape = once per engine, op and batch size: Op::launch_cache)
    const int even = cout_pad / 16 / 2 * 2;                         // tiles of the TPW = 2 launch; an odd last tile goes to a TPW = 1 launch
    if (off || cout_pad % 16 || (even != 2 && even != 4 && even != 8)) return false;
    if (stride == 2 && (cout_pad / 16 != even || getenv("ZLY_NO_WS_S2"))) return false;      // tuning / tests
    // 32 input channels at stride 2 are all patch DMA and no MFMA work: taken on maps up to 200k output pixels only (model.3 at 416 x 416: 23.2 -> 21.2 us;
    // YOLOv8-s' 320 -> 160 map at batch 32, 819k pixels: 68 -> 75 us)
    if (stride == 2 && cin == 32 && ((long)n * Ho * Wo >= 200000 || getenv("ZLY_NO_WS_S2_C32"))) return false;
    if (stride != 1 && stride != 2) return false;
    WsGeom g{};
    if (!ws_plan(Ho, Wo, cin, n, &g, stride)) return false;
    // enough pixels to give each of the 512 resident workgroups a ~13 x 13 tile (the tile planner would happily cut a small launch into
    // tiny tiles): below that the launch belongs to the latency-path kernels
    const double util = (double)Ho * Wo / ((double)g.tiles_x * g.tiles_y * g.TH * g.TW);
    const char* mt = getenv("ZLY_WS_MIN_TILES");
    if (util < 0.7 || (long)n * Ho * Wo < (mt ? atol(mt) : 64) * 169L) return false;     // 64 x 169 pixels (batch 16 at 26 x 26) up: batch 16 +2.8 %, batch 32 +6 % on one engine; 256 before
    cfg->ps = 1; cfg->ct = cout_pad / 16; cfg->pt = 4; cfg->ksplit = 1; cfg->fastk = 1;
    cfg->rowt = (stride == 1 && getenv("ZLY_WS_ROWT") != nullptr && g.TW + 2 <= 16) ? 1 : 0;      // experiment: one MFMA tile per output row, kx taps by DPP shifts
    {   // 64 -> 64 on small pixel tiles (the 26 x 26 maps: 7 x 13 tiles = 6 column tiles): 4 waves x ONE tile instead of 2 tile pairs x 2 pixel groups.  In the
        // pair form a wave there has three column tiles = two rounds of its software pipeline (MFMAs of a pair beside the epilogue of the one before), mostly
        // fill and drain: 6.7 k cycles for 1.7 k of MFMA issue (profiles/r03_ws_kernel_phase_stamps_v4.txt).  With one tile per wave it walks all six column
        // tiles (three rounds) for the same MFMAs: 12.5 -> 11.3 us per launch, step +1.9 % (nine launches).  On 13 x 13 tiles (11 column tiles) the pair form
        // wins (P3 box branch 23.3 vs 25.7 us: twice the fragment reads).  ZLY_WS_TPW1_MAXCT: tuning / tests (0 = never).
        const char* t1 = getenv("ZLY_WS_TPW1_MAXCT");
        cfg->tpw1 = (cin == 64 && cout_pad == 64 && (g.TH * g.TW + 15) / 16 <= (t1 ? atoi(t1) : 6)) ? 1 : 0;
    }
    return true;
}

// weight-stationary 1x1 kernel: single-source pointwise convs with at least four k-steps and enough pixels for the persistent grid
static bool pick_ws1_config(int cin, int cout_pad, int M, ConvLaunch* cfg, bool dual = false)
{
    Ws1Geom g{};
    int ny = 0;
    if (cin < 128 || !ws1_plan(cin, cout_pad, M, &g, &ny, dual)) return false;
    const char* mm = getenv("ZLY_WS1_MIN_PX");                      // tuning / tests: force the kernel onto small launches
    if (M < (mm ? atol(mm) : 2048)) return false;
    cfg->ws1 = 1; cfg->ct = cout_pad / 16; cfg->pt = g.npx / 16; cfg->ksplit = 1; cfg->fastk = getenv("ZLY_WS1_MAX_BYTES") ? 1 : 0; cfg->lds = 0; cfg->stream = 0;
    return true;
}

void conv_pick_config(int dtype, int ks, int stride, int cin, int cout_pad, int n, int Ho, int Wo, ConvLaunch* cfg, bool streamable, bool plain, bool dual)
{
    const int M = n * Ho * Wo;
    cfg->ks = ks; cfg->lds = 0; cfg->stream = 0; cfg->wres = 0; cfg->ps = 0; cfg->ws1 = 0; cfg->rowt = 0; cfg->tpw1 = 0;
    const bool no_stream = getenv("ZLY_NO_STREAM") != nullptr;             // tuning / tests
    // ZLY_WS1 (tuning / tests): 0 = never the weight-stationary 1x1 kernel, 1 = for the shapes the streaming kernel does not take, 2 = before it
    const char* w1 = getenv("ZLY_WS1");
    const int ws1_mode = w1 ? atoi(w1) : 1;
    const bool ws1_ok = dtype == ZLY_DTYPE_BF16 && ks == 1 && stride == 1 && streamable && ws1_mode > 0;
    if (dual) {             // Upsample + Concat input, otherwise "streamable": the weight-stationary kernel or the direct one
        if (dtype == ZLY_DTYPE_BF16 && ks == 1 && stride == 1 && ws1_mode > 0 && !getenv("ZLY_WS1_NO_DUAL") && pick_ws1_config(cin, cout_pad, M, cfg, true)) return;
        conv_pick_direct(dtype, ks, cin, cout_pad, M, cfg);
        return;
    }
    // before the streaming kernel: on request, and on the largest maps (YOLOv8-s 640 x 640 P3, 205k pixels: 31 -> 27 and 35 -> 31 us)
    if (ws1_ok && (ws1_mode == 2 || M >= 131072) && pick_ws1_config(cin, cout_pad, M, cfg)) return;
    if (dtype == ZLY_DTYPE_BF16 && ks == 1 && stride == 1 && streamable && !no_stream && pick_stream_config(cin, cout_pad, M, cfg)) return;
    if (ws1_ok && pick_ws1_config(cin, cout_pad, M, cfg)) return;
    if (dtype == ZLY_DTYPE_BF16 && ks == 3 && plain && pick_ws_config(stride, cin, cout_pad, n, Ho, Wo, cfg)) return;
    if (dtype == ZLY_DTYPE_BF16 && ks == 3 && pick_lds_config(stride, cin, cout_pad, n, Ho, Wo, cfg)) return;
    conv_pick_direct(dtype, ks, cin, cout_pad, M, cfg);
}

void conv_pick_direct(int dtype, int ks, int cin, int cout_pad, int M, ConvLaunch* cfg)
{
    const int kstep = conv_kstep(dtype);
    cfg->ks = ks; cfg->stream = 0; cfg->wres = 0; cfg->ps = 0; cfg->ws1 = 0; cfg->rowt = 0; cfg->tpw1 = 0;
    cfg->fastk = (ks == 3 && cin % kstep == 0) ? 1 : 0;
    cfg->ksplit = 1;
    const int ntiles = cout_pad / 16;
    static const int pref[5] = {4, 5, 3, 2, 1};
    cfg->ct = 1;
    for (int i = 0; i < 5; ++i)
        if (ntiles % pref[i] == 0) { cfg->ct = pref[i]; break; }
    cfg->lds = 0;
    if (dtype != ZLY_DTYPE_BF16) { cfg->pt = 2; return; }           // fp32 = verification mode: one shape
    const long ytiles = ntiles / cfg->ct;
    const long wgs_pt4 = ((long)(M + 255) / 256) * ytiles;
    const long wgs_pt2 = ((long)(M + 127) / 128) * ytiles;
    const long wgs_pt1 = ((long)(M + 63) / 64) * ytiles;
    const char* e4 = getenv("ZLY_DIRECT_PT4_MIN");                       // tuning aids
    const char* e2 = getenv("ZLY_DIRECT_PT2_MIN");
    const int nk = (ks * ks * cin + kstep - 1) / kstep;
    const char* fk = getenv("ZLY_DIRECT_KSPLIT_NK");                   // tuning aid: K-heavy launches take the 4-way split-K shape whatever their size
    const bool force_split = fk && nk >= atoi(fk);
    if (!force_split) {
        if (wgs_pt4 >= (e4 ? atol(e4) : 1024) && cfg->ct <= 4) { cfg->pt = 4; return; }       // CT=5 x PT=4 would need > 200 VGPRs
        if (wgs_pt2 >= (e2 ? atol(e2) : 512)) { cfg->pt = 2; return; }                      // 1024 before: the K-heavy 1x1 layers at 26x26 re-read their weights per 16 pixels (+0.8 %)
    }
    cfg->pt = 1;
    if (!force_split && (wgs_pt1 >= 512 || nk < 4)) return;
    cfg->ksplit = 4;                                                 // workgroup = one 16-pixel tile
    long wgs = ((long)(M + 15) / 16) * ytiles;
    while (wgs < 256 && cfg->ct > 1) {                               // still thin: fewer channel tiles per wave
        int nct = cfg->ct - 1;
        while (nct > 1 && ntiles % nct != 0) --nct;
        cfg->ct = nct;
        wgs = ((long)(M + 15) / 16) * (ntiles / cfg->ct);
    }
}

hipError_t launch_conv(int dtype, const ConvArgs& a, const ConvLaunch& cfg, hipStream_t s)
{
    if (cfg.ws1) {                                                 // weight-stationary 1x1 kernel
        Ws1Geom g{};
        int ny = 0;
        const bool dual = a.in2 != nullptr;
        if (dtype != ZLY_DTYPE_BF16 || cfg.ks != 1 || a.stride != 1 || a.pad != 0 || a.res || !a.act || a.out_f32 || a.Cin % 32 || a.nk != a.Cin / 32 ||
            a.Cout % 32 || a.cout_pad != a.Cout || a.in_cs % 8 || a.in_co % 8 || a.out_cs % 8 || a.out_co % 8 || !ws1_plan(a.Cin, a.cout_pad, a.M, &g, &ny, dual)) return hipErrorInvalidValue;
        if (dual && (a.in2_cs % 8 || a.in2_co % 8 || a.split_c % 8 || a.split_c <= 0 || a.split_c >= a.Cin || (a.H & 1) || (a.W & 1) || a.M % (a.H * a.W))) return hipErrorInvalidValue;
        // 32-bit byte offsets into the buffer resources: a tensor beyond 2 GiB (batch x map x channels far above any configuration run here)
        // takes the direct kernel's 64-bit addressing instead
        const size_t widest = (size_t)a.M * (size_t)std::max(std::max(a.in_cs, a.out_cs), dual ? a.in2_cs : 0) * 2;
        if (cfg.fastk || widest >= ((size_t)1 << 31)) {             // cfg.fastk: ZLY_WS1_MAX_BYTES was set when the shape was picked (tests: force the fall-back)
            ConvLaunch d{};
            conv_pick_direct(dtype, 1, a.Cin, a.cout_pad, a.M, &d);
            return launch_conv(dtype, a, d, s);
        }
        const size_t lds = ((size_t)g.npx * g.pitch + 1023) / 1024 * 1024;
        int gx = 2 * num_cus() / ny;                               // persistent: two resident workgroups per CU
        if (gx < 1) gx = 1;
        if (gx > g.total_tiles) gx = g.total_tiles;
        hipLaunchKernelGGL(pick_ws1(ws1_tpw(a.nk), a.nk, dual), dim3(gx, ny), dim3(256), lds, s, a, g);
        return hipGetLastError();
    }
    if (cfg.ps) {                                                  // weight-stationary 3x3 kernel
        WsGeom g{};
        const int ntiles = a.cout_pad / 16, even = ntiles / 2 * 2;
        if (dtype != ZLY_DTYPE_BF16 || !pick_ws(a.Cin, 2, false, false, a.stride) || a.pad != 1 || a.in2 || a.out_f32 || !a.act || a.nk != 9 * a.Cin / 32 ||
            a.in_cs % 8 || a.in_co % 8 || (even != 2 && even != 4 && even != 8) || !ws_plan(a.Ho, a.Wo, a.Cin, a.M / (a.Ho * a.Wo), &g, a.stride)) return hipErrorInvalidValue;
        if (a.stride == 2 && (a.res || ntiles != even)) return hipErrorInvalidValue;
        // 32-bit byte offsets into the buffer resources (and 0x80000000 as the out-of-range sentinel): a tensor of 2 GiB or more takes the LDS-tiled /
        // direct kernels' 64-bit addressing instead (YOLOv8-s 640 x 640 crosses it near batch 440).  ZLY_WS_MAX_BYTES forces it (tests).
        {
            const size_t widest = std::max(std::max((size_t)a.M / ((size_t)a.Ho * a.Wo) * a.H * a.W * (size_t)a.in_cs, (size_t)a.M * (size_t)a.out_cs), a.res ? (size_t)a.M * (size_t)a.res_cs : (size_t)0) * 2;
            const char* mb = getenv("ZLY_WS_MAX_BYTES");
            if (widest >= (mb ? (size_t)atoll(mb) : ((size_t)1 << 31))) {
                ConvLaunch d{};
                if (!pick_lds_config(a.stride, a.Cin, a.cout_pad, a.M / (a.Ho * a.Wo), a.Ho, a.Wo, &d)) conv_pick_direct(dtype, 3, a.Cin, a.cout_pad, a.M, &d);
                d.ks = 3;
                return launch_conv(dtype, a, d, s);
            }
        }
        const int n = a.M / (a.Ho * a.Wo);
        g.total_tiles = g.tiles_x * g.tiles_y * n;
        const size_t lds = a.stride == 2 ? ((size_t)(2 * g.TH + 1) * (g.TW + 8 * ((g.TW + 8) / 8)) * g.pitch + 1023) / 1024 * 1024
                                         : ((size_t)(g.TH + 2) * (g.TW + 8) * g.pitch + 1023) / 1024 * 1024;
        const int gx = g.total_tiles < 2 * num_cus() ? g.total_tiles : 2 * num_cus();      // persistent: two resident workgroups per CU
        // the even tiles: 4 waves = (even / 2) channel groups x pixel groups
        ConvArgs m = a;
        m.cout_pad = even * 16;
        m.Cout = a.Cout < even * 16 ? a.Cout : even * 16;
        g.nwc = even / 2; g.nwp = 4 / g.nwc;
        if (cfg.tpw1 && ntiles == 4 && a.Cin == 64) {
            // 64 output channels as 4 waves x ONE tile (every wave walks all column tiles of the pixel tile) instead of 2 tile pairs x 2 pixel groups: on
            // small pixel tiles a wave of the pair form has two or three column tiles -- one or two rounds of its software pipeline, mostly fill and drain
            g.nwc = 4; g.nwp = 1;
            if (a.stride == 2) hipLaunchKernelGGL(pick_ws(a.Cin, 1, false, false, 2), dim3(gx), dim3(256), lds, s, a, g);
            else               hipLaunchKernelGGL(pick_ws(a.Cin, 1, a.res != nullptr), dim3(gx), dim3(256), lds, s, a, g);
            return hipGetLastError();
        }
        if (a.stride == 2) {
            hipLaunchKernelGGL(pick_ws(a.Cin, 2, false, false, 2), dim3(gx), dim3(256), lds, s, m, g);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(pick_ws(a.Cin, 2, a.res != nullptr, cfg.rowt != 0 && a.Cin == 64 && g.TW + 2 <= 16), dim3(gx), dim3(256), lds, s, m, g);
        if (ntiles > even && a.Cout > even * 16) {
            // the odd last tile (pair-permuted rows cover the even tiles only, so it is a plain 16-channel conv of its own): 1 x 4 waves
            ConvArgs r = a;
            r.wgt = static_cast<const char*>(a.wgt) + (size_t)even * a.nk * 1024;
            r.bias = a.bias + even * 16;
            r.out_co = a.out_co + even * 16;
            if (a.res) r.res_co = a.res_co + even * 16;
            r.Cout = a.Cout - even * 16; r.cout_pad = 16;
            g.nwc = 1; g.nwp = 4;
            hipLaunchKernelGGL(pick_ws(a.Cin, 1, a.res != nullptr), dim3(gx), dim3(256), lds, s, r, g);
        }
        return hipGetLastError();
    }
    if (cfg.lds) {
        conv_lds_fn fn = pick_lds(a.stride, cfg.pt, cfg.ct);
        if (!fn || dtype != ZLY_DTYPE_BF16) return hipErrorInvalidValue;
        const int cout_pad = a.cout_pad;
        const int ytiles = cout_pad / (16 * cfg.ct);
        const int th = 4 * cfg.pt;
        const int tiles_x = (a.Wo + 15) / 16, tiles_y = (a.Ho + th - 1) / th;
        const int n = a.M / (a.Ho * a.Wo);
        const int tiles_per_img = tiles_x * tiles_y, total = tiles_per_img * n;
        int gx = total;
        const int nchunks = a.Cin / 32;
        if (cfg.wres && (a.stride != 1 || nchunks > 2)) return hipErrorInvalidValue;
        if (a.stride < 1 || a.stride > 2 || cfg.ct > 5 || cfg.pt > 4) return hipErrorInvalidValue;
        // resident workgroups per CU for this variant and LDS size, asked from the runtime once (registers: 94..243 per
        // lane, LDS 35..80 KB -> 2..4); the persistent grid is exactly what is resident
        static int occ_cache[3][6][5][3];                  // [stride][ct][pt][wres chunks] -> blocks per CU
        int& occ = occ_cache[a.stride][cfg.ct][cfg.pt][cfg.wres ? a.Cin / 32 : 0];
        if (occ == 0) {
            int o = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)fn, 256, lds_bytes(a.stride, cfg.pt, cfg.ct, cfg.wres ? a.Cin / 32 : 1)) != hipSuccess || o < 1) o = 2;
            occ = o > 4 ? 4 : o;
        }
        const char* wv = getenv("ZLY_LDS_WGS_PER_CU");       // tuning aid: force
        const int max_wgs = (wv ? atoi(wv) : occ) * num_cus();
        if (gx * ytiles > max_wgs) gx = max_wgs / ytiles;  // never more than are resident: a persistent workgroup
        if (gx > total) gx = total;                        // that has to wait for a slot runs a whole round alone
        hipLaunchKernelGGL(fn, dim3(gx, ytiles, 1), dim3(256), lds_bytes(a.stride, cfg.pt, cfg.ct, cfg.wres ? nchunks : 1), s, a, tiles_x, tiles_per_img, total, cfg.wres);
        return hipGetLastError();
    }
    if (cfg.ks == 1 && (a.stride != 1 || a.pad != 0)) return hipErrorInvalidValue;      // the 1x1 paths assume input pixel = output pixel
    if (cfg.stream) {
        const int nk = (a.Cin + 31) / 32;
        conv_stream_fn sf = pick_stream(cfg.ct, cfg.pt, nk);
        if (!sf || dtype != ZLY_DTYPE_BF16 || a.in2 || a.res || nk > a.nk || !a.act || a.out_f32 || a.Cout % 32 || a.in_cs % 8 || a.in_co % 8 || a.out_cs % 8 || a.out_co % 8)
            return hipErrorInvalidValue;
        const int cout_pad = a.cout_pad;
        const int ytiles = cout_pad / (16 * cfg.ct);
        const int ngroups = (a.M + 16 * cfg.pt - 1) / (16 * cfg.pt);
        int gx = (ngroups + 3) / 4;
        const char* sw = getenv("ZLY_STREAM_WGS");         // tuning / tests: total persistent workgroups (default ~4 per CU)
        const int wgs = sw && atoi(sw) > 0 ? atoi(sw) : (cfg.ct == 4 && cfg.pt == 1 ? 2 : 4) * num_cus();      // the 64-channel shapes hold 2 workgroups per CU (189-239 VGPRs)
        const int cap = wgs / ytiles > 0 ? wgs / ytiles : 1;
        if (gx > cap) gx = cap;
        hipLaunchKernelGGL(sf, dim3(gx, ytiles, 1), dim3(256), 0, s, a, ngroups);
        return hipGetLastError();
    }
    const int mode = cfg.ks == 1 ? 0 : (cfg.fastk ? 1 : 2);
    conv_fn fn = nullptr;
    // the split-K kernel reads its inputs through 32-bit buffer offsets: an input tensor of 2 GiB or more (never a latency-path launch) takes the one-pass shape
    const size_t in_bytes = (size_t)(a.M / (a.Ho * a.Wo)) * a.H * a.W * (size_t)std::max(a.in_cs, a.in2 ? a.in2_cs : 0) * 2;
    const bool split = cfg.ksplit == 4 && in_bytes < ((size_t)1 << 31) && a.M < (1 << 20) && a.Wo < 1024 && a.Ho < 1024;
    if (dtype == ZLY_DTYPE_BF16) {
        if (split)             fn = pick_mode<bf16_t, 1, 4>(mode, cfg.ct);
        else if (cfg.pt == 1)  fn = pick_mode<bf16_t, 1, 1>(mode, cfg.ct);
        else if (cfg.pt == 2)  fn = pick_mode<bf16_t, 2, 1>(mode, cfg.ct);
        else                   fn = pick_mode<bf16_t, 4, 1>(mode, cfg.ct);
    } else {
        fn = pick_mode<float, 2, 1>(mode, cfg.ct);
    }
    if (!fn) return hipErrorInvalidValue;
    const int cout_pad = a.cout_pad;
    const int ytiles = cout_pad / (16 * cfg.ct);
    const int px_per_wg = split ? 16 * cfg.pt : 64 * cfg.pt;
    dim3 grid((a.M + px_per_wg - 1) / px_per_wg, ytiles, 1);
    ConvArgs b = a;
    b.inv_wo = 1.0f / (float)a.Wo; b.inv_ho = 1.0f / (float)a.Ho;
    hipLaunchKernelGGL(fn, grid, dim3(256), 0, s, b);
    return hipGetLastError();
}

}  // namespace zly
