// kernels_c2f64.hip -- a 64-channel C2f block (the 26 x 26 stage of YOLOv8n at 416 x 416: model.6, model.12, model.18) as ONE kernel, bf16.
//
// Replaces the 4 (one bottleneck) or 6 (two) conv nodes of a C2f that the reference executes inside Ort::Session::Run
// (reference src/inference/onnx_engine.cpp:578-585): ultralytics C2f = cv2(cat(chunk(cv1(x), 2), m_0(y1), m_1(y2) ...)), m_i = 3x3 -> 3x3 (+ x).
//
// Why: at this stage a conv is 1.3-2.7 us of work behind a 10-18 us launch (dispatch + drain + every workgroup loading the layer's weights for
// ONE pixel tile); profiles/r03_per_launch.json: model.12 = 62 us in 4 launches, model.18 = 59 us, model.6 = 88 us in 6.  The block has no
// reason to leave the CU: a 13 x 13 output tile needs a 17 x 17 patch of cv1's output (two 3x3 convs: halo 2), 46 KB at 64 channels.
//
//   MODE 3  cv1 -> bottleneck -> cv2          a whole C2f with one bottleneck (model.12, model.18)
//   MODE 1  cv1 -> bottleneck                 front half of a C2f with two bottlenecks (model.6): y0 | y1 | y2 go to the concat buffer in HBM
//   MODE 2  bottleneck -> cv2                 back half: the patch (y2) is staged from the concat buffer, y0 | y1 are read from it by cv2
//
// Same structure as c2f_kernel (kernels_pair.hip) for 16 / 32 channels, but the weights (cv1 32-96 KB, 74 KB per 3x3, cv2 48-64 KB) do not fit
// LDS next to the maps, so every phase gets them its own way:
//   A  cv1 (1x1, Cin -> 128) on every pixel of the (TH+4) x (TW+4) patch: weights stream through LDS in chunks of NKC1 k-steps (LDS-DMA into
//      the space of the maps that are not live yet), pixel fragments straight from global memory (dual source: the neck's fused Upsample +
//      Concat); a wave owns up to 3 linearised 16-pixel tiles x all 8 channel tiles.  y1 -> patch buffer, y0 (tile interior) -> LDS map.
//   B  first 3x3 on the (TH+2) x (TW+2) region, C  second 3x3 (+ shortcut) on TH x TW: WEIGHT-STATIONARY as conv3x3_ws_kernel -- a wave keeps
//      the 18 k-steps of 2 channel tiles in registers (144 VGPRs, loaded from L2 at the start of the phase), 2 channel groups x 4 pixel groups
//      = 8 waves; a 16-pixel MFMA tile is one ROW of the region (TW + 2 <= 16), so fragment reads never wrap a row: conflict-free at the
//      160-byte pixel pitch with minimal row pitches.  Zero outside the frame where the next conv pads.
//   D  cv2 (1x1, 64 x (2 + n) -> 128) over the concat [y0 | y1 | y2 (| y3)] taken from the LDS maps (MODE 2: y0 | y1 from HBM), weights through
//      LDS in chunks of NKC2 k-steps (into the first 3x3's dead intermediate map), a wave owns up to 2 pixel tiles x all 8 channel tiles.
// Every intermediate is rounded to bf16 exactly where the unfused path rounds it (bias + SiLU in fp32, then bf16): results equal the
// one-kernel-per-conv path up to fp32 summation order.  One workgroup of 8 waves per CU (136 KB of LDS at 13 x 13).
#include "zly_internal.h"
#include "conv_device.h"
#include <stdlib.h>
#include <stdio.h>

namespace zly {

static constexpr int C64_PITCH = 160;          // bytes per map pixel: 128 + 32 (conflict-free ds_read_b128 of 16 consecutive pixels, tools/lds_pitch.py)
static constexpr int C64_NW = 8;
static constexpr int C64_LDS_MAX = 160 * 1024;

struct C64Layout { int p, y2, m, total; };
// LDS regions, each 1 KiB aligned (LDS-DMA destinations): [patch | y map (modes with cv2) | intermediate map | extra]; cv1's weights (8 x nk1 KiB)
// live in [y map .. extra) (dead during phase A), cv2's (8 x nk2 KiB) in [intermediate map .. extra) (dead during phase D)
__host__ __device__ inline C64Layout c64_layout(int mode, int th, int tw, int nk1, int nk2)
{
#define C64_UP(b) (((b) + 1023) / 1024 * 1024)
    C64Layout L;
    L.p = 0;
    L.y2 = C64_UP((th + 4) * (tw + 4) * C64_PITCH);
    L.m = L.y2 + ((mode & 2) ? C64_UP(th * tw * C64_PITCH) : 0);
    int end = L.m + C64_UP((th + 2) * (tw + 2) * C64_PITCH);
#undef C64_UP
    if ((mode & 1) && end - L.y2 < nk1 * 8192) end = L.y2 + nk1 * 8192;
    if ((mode & 2) && end - L.m < nk2 * 8192) end = L.m + nk2 * 8192;
    L.total = end;
    return L;
}

// 16-byte accesses through buffer resources: 32-bit byte offsets in ONE VGPR (+ an SGPR), so that the compiler has no 64-bit address pairs to
// precompute, hoist out of the tile loop and spill (the first build kept 72 weight addresses = 144 VGPRs live across the whole kernel)
__device__ __forceinline__ bf16x8 c64_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
__device__ __forceinline__ void c64_st8(__amdgpu_buffer_rsrc_t r, int voff, f32x4 a, f32x4 b)
{
    const bf16x8 o = to_bf16x8(a, b);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r, voff, 0, 0);
}

__device__ __forceinline__ int c64_div(int q, float inv) { return (int)(((float)q + 0.5f) * inv); }     // exact for q < 2^20, divisor < 2^10

// `n` k-steps [s0, s0 + n) of an 8-tile 1x1 weight array ([tile][nk][lane][8]) -> LDS as [tile][n][lane][8], by LDS-DMA (1 KiB per wave instruction)
__device__ __forceinline__ void c64_dma_weights(const void* w, int nk, int s0, int n, unsigned char* dst, int wave, int lane, unsigned total_bytes)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, total_bytes, 0x00020000);
    for (int k = wave; k < 8 * n; k += C64_NW) {
        const int c = k / n, j = k - c * n;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + k * 1024), 16, (unsigned)(((c * nk + s0 + j) * 1024) + lane * 16), 0, 0, 0);
    }
}

// acc[c] += W[c][j] x X[j] over NK k-steps and the 8 channel tiles of a 1x1 conv: weight fragments from LDS ([tile][NK][lane][8], wl = base + lane * 16),
// read one k-step ahead of the MFMAs that use them; the scheduling barriers keep the compiler from issuing all 8 x NK reads up front (it
// did, and spilled the pixel fragments to make room)
template <int NK>
__device__ __forceinline__ void c64_gemm8(const unsigned char* wl, const bf16x8 (&x)[NK], f32x4 (&acc)[8])
{
    bf16x8 wf[2][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) wf[0][c] = *reinterpret_cast<const bf16x8*>(wl + (c * NK) * 1024);
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        if (j + 1 < NK) {
#pragma unroll
            for (int c = 0; c < 8; ++c) wf[(j + 1) & 1][c] = *reinterpret_cast<const bf16x8*>(wl + (c * NK + j + 1) * 1024);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = mma_step(wf[j & 1][c], x[j], acc[c]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One 3x3 conv phase, weight-stationary: rows [0, nrows) x columns [0, ncols) of the output region; source map `src` with `spw` pixels per
// row (output (r, c) reads source pixels (r + ky, c + kx)); this wave: channel tiles 2 wc, 2 wc + 1 (weights w), rows wp, wp + 4, ...
// two rows at a time.  epi(row, col, acc[2]) is called for the lanes col < ncols of every valid row.
template <typename Epi>
__device__ __forceinline__ void c64_conv3x3(const unsigned char* src, int spw, int nrows, int ncols, const bf16x8 (&w)[2][18], int wp, int p, int kq, Epi&& epi)
{
    int toff[18];
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        const int tap = s >> 1, chunk = s & 1;
        const int ky = tap / 3, kx = tap - ky * 3;
        toff[s] = (ky * spw + kx) * C64_PITCH + chunk * 64;
    }
    const int col = min(p, ncols - 1);
    for (int r0 = wp; r0 < nrows; r0 += 8) {
        const unsigned char* px[2];
        int rr[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            rr[h] = r0 + 4 * h;
            px[h] = src + (min(rr[h], nrows - 1) * spw + col) * C64_PITCH + kq * 16;
        }
        f32x4 acc[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) { acc[h][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[h][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        constexpr int DEPTH = 3;
        bf16x8 xf[2][DEPTH + 1];
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) { xf[0][s] = *reinterpret_cast<const bf16x8*>(px[0] + toff[s]); xf[1][s] = *reinterpret_cast<const bf16x8*>(px[1] + toff[s]); }
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            if (s + DEPTH < 18) {
                xf[0][(s + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(px[0] + toff[s + DEPTH]);
                xf[1][(s + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(px[1] + toff[s + DEPTH]);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc[0][c] = mma_step(w[c][s], xf[0][s % (DEPTH + 1)], acc[0][c]);
                acc[1][c] = mma_step(w[c][s], xf[1][s % (DEPTH + 1)], acc[1][c]);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (rr[h] < nrows && p < ncols) epi(rr[h], p, acc[h]);
    }
}

#ifdef ZLY_C64_DIAG
__device__ unsigned long long* g_c64_diag = nullptr;             // diagnostic build only (tools/c64_bench.hip): per-wave cycle sums of the phases
#define C64STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); dsum[k] += t_ - dT0; dT0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define C64STAMP(k) do { } while (0)
#endif

template <int MODE, int NK1, int NK2>
__global__ __launch_bounds__(C64_NW * 64) void c2f64_kernel(const C2fArgs a)
{
#ifdef ZLY_C64_DIAG
    unsigned long long dsum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dT0 = __builtin_amdgcn_s_memtime();
    const unsigned long long dstart = dT0;
#endif
    constexpr bool FRONT = (MODE & 1) != 0, BACK = (MODE & 2) != 0;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int TH = a.TH, TW = a.TW, PH = TH + 4, PW = TW + 4, MH = TH + 2, MW = TW + 2;
    const C64Layout L = c64_layout(MODE, TH, TW, NK1, NK2);
    unsigned char* lP = smem + L.p;
    unsigned char* lY2 = smem + L.y2;
    unsigned char* lM = smem + L.m;
    unsigned char* lW1 = smem + L.y2;            // cv1's weights (phase A)
    unsigned char* lW2 = smem + L.m;             // cv2's weights (phase D)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wp = wave >> 1;      // 3x3 phases: channel group (tiles 2 wc, 2 wc + 1 = channels wc*32 .. +31), pixel-row group
    const unsigned map_px = (unsigned)a.n * a.H * a.W;
    const __amdgpu_buffer_rsrc_t rcat = __builtin_amdgcn_make_buffer_rsrc(a.cat, 0, map_px * a.cat_cs * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wA), 0, 4 * 18 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wB), 0, 4 * 18 * 1024, 0x00020000);
    const float invPW = 1.0f / (float)PW, invTW = 1.0f / (float)TW;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int NP0 = PH * PW, NPB = TH * TW;
    const int nt0 = (NP0 + 15) >> 4, ntB = (NPB + 15) >> 4;

    for (int tl = blockIdx.x; tl < a.total_tiles; tl += gridDim.x) {
        // the lane's coordinates are made opaque per tile: everything derived from them (LDS addresses of four phases, global offsets) is
        // tile-invariant, and the compiler otherwise computes all of it before the loop and carries it through every phase in scratch
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        const int p = lane_v & 15, kq = lane_v >> 4;
        const int b = tl / tiles_per_img;
        const int rt = tl - b * tiles_per_img;
        const int ty = rt / a.tiles_x;
        const int y0 = ty * TH, x0 = (rt - ty * a.tiles_x) * TW;

        if (FRONT) {
            // ---- A. cv1 on every pixel of the patch: x (global) -> y0 (concat buffer in HBM, tile interior) | y1 (patch buffer) ---------
            const bool dual = a.x2 != nullptr;
            const int split = dual ? a.split_c : (1 << 30);
            const __amdgpu_buffer_rsrc_t rxa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (dual ? map_px / 4 : map_px) * a.x_cs * 2, 0x00020000);
            const __amdgpu_buffer_rsrc_t rxb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(dual ? a.x2 : a.x), 0, map_px * (dual ? a.x2_cs : a.x_cs) * 2, 0x00020000);
            // pixel tile t of the patch (linearised): this lane's pixel, its fragments for all NK1 k-steps
            auto load_x = [&](int t, bf16x8 (&xf)[NK1]) {
                const int qc = min(t * 16 + p, NP0 - 1);
                const int py = c64_div(qc, invPW), px = qc - py * PW;
                const int gy = min(max(y0 - 2 + py, 0), a.H - 1), gx = min(max(x0 - 2 + px, 0), a.W - 1);      // outside the frame: any valid pixel (zeroed below)
                const int pa = (dual ? ((b * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1)) * a.x_cs : ((b * a.H + gy) * a.W + gx) * a.x_cs) * 2 + (a.x_co + kq * 8) * 2;
                const int pb2 = ((b * a.H + gy) * a.W + gx) * a.x2_cs * 2 + (a.x2_co + kq * 8) * 2;
#pragma unroll
                for (int j = 0; j < NK1; ++j) {
                    const int ci = j * 32;                                  // first channel of the k-step (split_c is a multiple of 32)
                    const bool second = ci >= split;                        // wave-uniform: selects, not branches (a branch per fragment serialised the loads)
                    xf[j] = c64_ld(second ? rxb : rxa, second ? pb2 : pa, second ? (ci - split) * 2 : ci * 2);
                }
            };
            __syncthreads();                                     // the previous tile's readers of the maps under the weight region are done
            c64_dma_weights(a.w1, NK1, 0, NK1, lW1, wave, lane_v, (unsigned)(8 * NK1 * 1024));
            bf16x8 xA[NK1], xB[NK1];                             // ping-pong: the next tile's fragments are in flight while this one computes
            if (wave < nt0) load_x(wave, xA);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // cv1's weights are in LDS
            C64STAMP(0);
            const unsigned char* wl = lW1 + lane_v * 16;
            auto round_a = [&](int t, const bf16x8 (&xcur)[NK1], bf16x8 (&xnext)[NK1]) {
                asm volatile("" ::: "memory");                   // keep the weight fragments in LDS: left alone the compiler hoists the loop-invariant reads (8 x NK1 x 4 registers) and spills
                if (t + C64_NW < nt0) load_x(t + C64_NW, xnext);
                f32x4 acc[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                c64_gemm8<NK1>(wl, xcur, acc);
                const int q = t * 16 + p;
                const int qc = min(q, NP0 - 1);
                const int py = c64_div(qc, invPW), px = qc - py * PW;
                const int gy = y0 - 2 + py, gx = x0 - 2 + px;
                const bool inimg = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const bool interior = (unsigned)(py - 2) < (unsigned)TH && (unsigned)(px - 2) < (unsigned)TW && inimg;
#pragma unroll
                for (int g2 = 0; g2 < 4; ++g2) {                 // channel pairs: g2 = 0, 1 -> y0 channels g2*32 + kq*8 .. +7; 2, 3 -> y1
                    f32x4 lo = acc[2 * g2] + *reinterpret_cast<const f32x4*>(a.b1 + g2 * 32 + kq * 8);
                    f32x4 hi = acc[2 * g2 + 1] + *reinterpret_cast<const f32x4*>(a.b1 + g2 * 32 + kq * 8 + 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { lo[r] = inimg ? silu<bf16_t>(lo[r]) : 0.0f; hi[r] = inimg ? silu<bf16_t>(hi[r]) : 0.0f; }      // outside the frame: the 3x3 convs' zero padding
                    if (q < NP0) {
                        if (g2 >= 2) store8(reinterpret_cast<bf16_t*>(lP + (size_t)qc * C64_PITCH) + (g2 - 2) * 32 + kq * 8, lo, hi);
                        // y0 is only consumed by cv2: it goes to its place in the concat buffer (L2) and comes back in phase D; y1 too in MODE 1
                        if (interior && (g2 < 2 || MODE == 1 || a.dump))
                            c64_st8(rcat, (((b * a.H + gy) * a.W + gx) * a.cat_cs + g2 * 32 + kq * 8) * 2, lo, hi);
                    }
                }
            };
            for (int t = wave; t < nt0; t += 2 * C64_NW) {
                round_a(t, xA, xB);
                if (t + C64_NW < nt0) round_a(t + C64_NW, xB, xA);
            }
            C64STAMP(1);
        } else {
            // ---- MODE 2: the bottleneck's input patch from the concat buffer, by LDS-DMA; outside the frame / pitch padding: out-of-range -> 0 ----
            __syncthreads();                                     // previous tile's readers of the patch are done
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(static_cast<bf16_t*>(a.cat) + a.pair_in_co, 0, (map_px * a.cat_cs - a.pair_in_co) * 2, 0x00020000);
            const int NLU = NP0 * (C64_PITCH / 16), ndma = (NLU + 63) >> 6;
            for (int k = wave; k < ndma; k += C64_NW) {
                const int u = k * 64 + lane_v;
                const int px = (int)(((float)u + 0.5f) * 0.1f), part = u - px * 10;
                const int py = c64_div(px, invPW), pxx = px - py * PW;
                const int gy = y0 - 2 + py, gx = x0 - 2 + pxx;
                const bool ok = part < 8 && py < PH && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const unsigned off = ok ? (unsigned)((((b * a.H + gy) * a.W + gx) * a.cat_cs) * 2 + part * 16) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(lP + k * 1024), 16, off, 0, 0, 0);
            }
        }

        // ---- B. first 3x3: patch -> intermediate map (zero outside the frame) -----------------------------------------------------------
        bf16x8 w[2][18];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 18; ++s) w[t][s] = c64_ld(rwA, lane_v * 16, ((wc * 2 + t) * 18 + s) * 1024);
        {
            const f32x4 blo = *reinterpret_cast<const f32x4*>(a.bA + wc * 32 + kq * 8), bhi = *reinterpret_cast<const f32x4*>(a.bA + wc * 32 + kq * 8 + 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // MODE 2: this wave's pieces of the patch have landed
            C64STAMP(2);
            __syncthreads();                                     // patch complete
            C64STAMP(3);
            c64_conv3x3(lP, PW, MH, MW, w, wp, p, kq, [&](int my, int mx, const f32x4 (&acc)[2]) {
                const int gy = y0 - 1 + my, gx = x0 - 1 + mx;
                const bool inimg = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                f32x4 lo = acc[0] + blo, hi = acc[1] + bhi;
#pragma unroll
                for (int r = 0; r < 4; ++r) { lo[r] = inimg ? silu<bf16_t>(lo[r]) : 0.0f; hi[r] = inimg ? silu<bf16_t>(hi[r]) : 0.0f; }
                store8(reinterpret_cast<bf16_t*>(lM + (size_t)(my * MW + mx) * C64_PITCH) + wc * 32 + kq * 8, lo, hi);
                if (a.dump && a.mid && inimg && (unsigned)(my - 1) < (unsigned)TH && (unsigned)(mx - 1) < (unsigned)TW)       // debug tap: the pixels this tile owns
                    store8(static_cast<bf16_t*>(a.mid) + ((size_t)(b * a.H + gy) * a.W + gx) * a.mid_cs + wc * 32 + kq * 8, lo, hi);
            });
        }

        // ---- C. second 3x3 (+ shortcut from the patch): intermediate -> y ------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 18; ++s) w[t][s] = c64_ld(rwB, lane_v * 16, ((wc * 2 + t) * 18 + s) * 1024);
        {
            const f32x4 blo = *reinterpret_cast<const f32x4*>(a.bB + wc * 32 + kq * 8), bhi = *reinterpret_cast<const f32x4*>(a.bB + wc * 32 + kq * 8 + 4);
            C64STAMP(4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            C64STAMP(5);
            __syncthreads();                                     // intermediate map complete
            C64STAMP(6);
            c64_conv3x3(lM, MW, TH, TW, w, wp, p, kq, [&](int oy, int ox, const f32x4 (&acc)[2]) {
                f32x4 lo = acc[0] + blo, hi = acc[1] + bhi;
#pragma unroll
                for (int r = 0; r < 4; ++r) { lo[r] = silu<bf16_t>(lo[r]); hi[r] = silu<bf16_t>(hi[r]); }
                if (a.res) {
                    f32x4 ra, rb;
                    load8(reinterpret_cast<const bf16_t*>(lP + (size_t)((oy + 2) * PW + ox + 2) * C64_PITCH) + wc * 32 + kq * 8, ra, rb);
                    lo += ra; hi += rb;
                }
                const int gy = y0 + oy, gx = x0 + ox;
                if (BACK) store8(reinterpret_cast<bf16_t*>(lY2 + (size_t)(oy * TW + ox) * C64_PITCH) + wc * 32 + kq * 8, lo, hi);
                if ((!BACK || a.dump) && gy < a.H && gx < a.W)
                    c64_st8(rcat, (((b * a.H + gy) * a.W + gx) * a.cat_cs + a.pair_out_co + wc * 32 + kq * 8) * 2, lo, hi);
            });
        }

        C64STAMP(7);
        if (BACK) {
            // ---- D. cv2 over the concat [y0 | y1 | y (| y3)] -> out: y0 (MODE 2: y0 | y1) from the concat buffer, the rest from the LDS maps ---
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, map_px * a.out_cs * 2, 0x00020000);
            constexpr int NMAPS = NK2 / 2, NGLOB = MODE == 2 ? 2 : 1;        // source maps of 64 channels; how many of them come from HBM
            auto load_x2 = [&](int t, bf16x8 (&xf)[NK2], int& oy, int& ox) {
                const int qc = min(t * 16 + p, NPB - 1);
                oy = c64_div(qc, invTW); ox = qc - oy * TW;
                const int gp = (((b * a.H + min(y0 + oy, a.H - 1)) * a.W + min(x0 + ox, a.W - 1)) * a.cat_cs + kq * 8) * 2;
#pragma unroll
                for (int j = 0; j < NK2; ++j) {
                    const int map = j >> 1, half = j & 1;
                    if (map < NGLOB) xf[j] = c64_ld(rcat, gp, (map * 64 + half * 32) * 2);
                    else if (map == NMAPS - 2) xf[j] = *reinterpret_cast<const bf16x8*>(lP + (size_t)((oy + 2) * PW + ox + 2) * C64_PITCH + half * 64 + kq * 16);
                    else xf[j] = *reinterpret_cast<const bf16x8*>(lY2 + (size_t)qc * C64_PITCH + half * 64 + kq * 16);
                }
            };
            __syncthreads();                                     // y map complete (and y0 in L2: the barrier waits for this wave's stores), intermediate map dead
            c64_dma_weights(a.w2, NK2, 0, NK2, lW2, wave, lane_v, (unsigned)(8 * NK2 * 1024));
            bf16x8 xA[NK2], xB[NK2];
            int oyA = 0, oxA = 0, oyB = 0, oxB = 0;
            if (wave < ntB) load_x2(wave, xA, oyA, oxA);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // cv2's weights are in LDS
            C64STAMP(8);
            const unsigned char* wl = lW2 + lane_v * 16;
            auto round_d = [&](int t, const bf16x8 (&xcur)[NK2], int oy, int ox, bf16x8 (&xnext)[NK2], int& oyn, int& oxn) {
                asm volatile("" ::: "memory");                   // as in phase A
                if (t + C64_NW < ntB) load_x2(t + C64_NW, xnext, oyn, oxn);
                f32x4 acc[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                c64_gemm8<NK2>(wl, xcur, acc);
                const int q = t * 16 + p;
                if (q < NPB && y0 + oy < a.H && x0 + ox < a.W) {
                    const int dst = (((b * a.H + y0 + oy) * a.W + x0 + ox) * a.out_cs + a.out_co + kq * 8) * 2;
#pragma unroll
                    for (int g2 = 0; g2 < 4; ++g2) {
                        f32x4 lo = acc[2 * g2] + *reinterpret_cast<const f32x4*>(a.b2 + g2 * 32 + kq * 8);
                        f32x4 hi = acc[2 * g2 + 1] + *reinterpret_cast<const f32x4*>(a.b2 + g2 * 32 + kq * 8 + 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { lo[r] = silu<bf16_t>(lo[r]); hi[r] = silu<bf16_t>(hi[r]); }
                        c64_st8(rout, dst + g2 * 64, lo, hi);
                    }
                }
            };
            for (int t = wave; t < ntB; t += 2 * C64_NW) {
                round_d(t, xA, oyA, oxA, xB, oyB, oxB);
                if (t + C64_NW < ntB) round_d(t + C64_NW, xB, oyB, oxB, xA, oyA, oxA);
            }
        }
        C64STAMP(9);
        // the next tile's first barrier (phase A's, or MODE 2's before the patch DMA) orders this tile's LDS reads before its writes
    }
#ifdef ZLY_C64_DIAG
    if (lane == 0 && g_c64_diag) {
        unsigned long long* o = g_c64_diag + ((size_t)blockIdx.x * C64_NW + wave) * 16;
        for (int k = 0; k < 10; ++k) o[k] = dsum[k];
        o[10] = __builtin_amdgcn_s_memtime() - dstart;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
typedef void (*c2f64_fn)(const C2fArgs);
struct C64Variant { int mode, nk1, nk2; c2f64_fn fn; };
// YOLOv8n: model.12 (384 -> 128, one bottleneck), model.18 (192 -> 128, one), model.6 front (128 -> 128) + back (two bottlenecks: concat of 4)
static const C64Variant c64_variants[] = {
    {3, 12, 6, c2f64_kernel<3, 12, 6>},
    {3, 6, 6, c2f64_kernel<3, 6, 6>},
    {3, 4, 6, c2f64_kernel<3, 4, 6>},
    {1, 4, 0, c2f64_kernel<1, 4, 1>},
    {1, 6, 0, c2f64_kernel<1, 6, 1>},
    {2, 0, 8, c2f64_kernel<2, 1, 8>},
};
static const C64Variant* c64_pick(int mode, int nk1, int nk2)
{
    for (const C64Variant& v : c64_variants)
        if (v.mode == mode && ((mode & 1) == 0 || v.nk1 == nk1) && ((mode & 2) == 0 || v.nk2 == nk2)) return &v;
    return nullptr;
}

// tile shape: TW + 2 <= 16 (a region row is one MFMA pixel tile), phase A <= 3 and phase D <= 2 pixel tiles per wave, LDS budget; among those
// the least (rounds of tiles over the CUs) x (MFMAs on the busiest wave per tile + a fixed per-tile cost: weight loads, barriers)
bool c2f64_plan(int mode, int nk1, int nk2, int cout2, int n, int H, int W, C2fPlan* plan)
{
    const C64Variant* v = c64_pick(mode, nk1, nk2);
    // OFF unless ZLY_C2F64 is set: measured (profiles/r03_c2f64_*): model.12 62 -> 57 us, model.18 59 -> 42, model.6 88 -> 56 at batch 64, but
    // the step with three engines gets SLOWER (0.676 -> 0.695 ms): one 121-142 KB workgroup per CU for 40-57 us keeps the other chains' kernels
    // off the CU, which the four short launches it replaces did not; batch 1: 0.205 -> 0.225 ms (35 workgroups of 4 x 6 pixels, 17 us each)
    if (!v || ((mode & 2) && cout2 != 128) || !getenv("ZLY_C2F64")) return false;
    const int ncu = num_cus();
    double best = 1e30;
    for (int th = 4; th <= 16; ++th)
        for (int tw = 6; tw <= 14; ++tw) {
            const C64Layout L = c64_layout(mode, th, tw, v->nk1 ? v->nk1 : 1, v->nk2 ? v->nk2 : 1);
            if (L.total > C64_LDS_MAX) continue;
            const int tx = (W + tw - 1) / tw, ty = (H + th - 1) / th;
            const long tiles = (long)n * tx * ty;
            const long rounds = (tiles + ncu - 1) / ncu;
            const int nt0 = ((th + 4) * (tw + 4) + 15) / 16, ntB = (th * tw + 15) / 16;
            double per_tile = 400.0 + 36.0 * ((th + 2 + 3) / 4) + 36.0 * ((th + 3) / 4);
            if (mode & 1) per_tile += 8.0 * nk1 * ((nt0 + C64_NW - 1) / C64_NW);
            if (mode & 2) per_tile += 8.0 * nk2 * ((ntB + C64_NW - 1) / C64_NW);
            const double cost = (double)rounds * per_tile;
            if (cost < best) { best = cost; plan->th = th; plan->tw = tw; plan->tiles_x = tx; plan->tiles_y = ty; plan->total_tiles = (int)tiles; plan->lds_bytes = L.total; }
        }
    if (best >= 1e30) return false;
    plan->grid = plan->total_tiles < ncu ? plan->total_tiles : ncu;
    return true;
}

hipError_t c2f64_init()
{
    for (const C64Variant& v : c64_variants) {
        hipError_t r = hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS_MAX);
        if (r != hipSuccess) return r;
    }
    return hipSuccess;
}

hipError_t launch_c2f64(int mode, const C2fArgs& a, const C2fPlan& plan, hipStream_t s)
{
    const C64Variant* v = c64_pick(mode, a.nk1, a.nk2);
    int why = 0;
    if (!v || a.TH != plan.th || a.TW != plan.tw || plan.grid < 1 || a.TW + 2 > 16) why = 1;
    else if (a.cat_cs % 8 || a.pair_in_co % 64 || a.pair_out_co % 64 || a.out_cs % 8 || a.out_co % 8) why = 2;
    else if ((mode & 1) && (a.x_cs % 8 || a.x_co % 8 || (a.x2 && (a.x2_cs % 8 || a.x2_co % 8 || a.split_c % 32 || (a.H & 1) || (a.W & 1))))) why = 3;
    else if ((mode & 2) && a.Cout2 != 128) why = 4;
    else if (mode == 2 && a.pair_in_co != 128) why = 5;
    else if (c64_layout(mode, a.TH, a.TW, v->nk1 ? v->nk1 : 1, v->nk2 ? v->nk2 : 1).total != plan.lds_bytes) why = 6;
    if (why) {
        fprintf(stderr, "zly: launch_c2f64 rejected its arguments (check %d: mode %d nk1 %d nk2 %d tile %dx%d map %dx%d cat_cs %d in_co %d out_co %d)\n", why, mode, a.nk1, a.nk2,
                a.TH, a.TW, a.H, a.W, a.cat_cs, a.pair_in_co, a.pair_out_co);
        return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(v->fn, dim3((unsigned)plan.grid), dim3(C64_NW * 64), (size_t)plan.lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace zly
