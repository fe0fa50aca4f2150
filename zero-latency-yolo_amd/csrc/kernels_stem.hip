// kernels_stem.hip -- preprocess fused into the YOLOv8 stem conv (model.0: 3->c0, k3 s2 p1 + SiLU), bf16.
//
// Replaces two launches and the HBM round trip of the preprocessed tensor: the reference materialises
// preProcess's fp32 [3][416][416] tensor (onnx_engine.cpp:649-700) and hands it to the first ORT conv node.
// Here a workgroup owns a 16 x 32 tile of stem outputs of one frame (8 x 32 at first: the per-workgroup setup -- three
// IEEE divides, weight fragments, tap offsets -- was a fifth of a wave's instructions in this issue-bound kernel):
//   1. pixel values become bf16(u8 * (1/255.f)), which equals bf16(u8 / 255.0f) -- what preProcess + bf16 rounding gives
//      (:693) -- for every u8 (checked exhaustively on the CPU), so no divide and no lookup table;
//   2. stages the 33 x 65 input patch: for every model-space pixel the reference's nearest-neighbour map
//      src = (min(int(y*scale_h), h-1), min(int(x*scale_w), w-1)) (:673-685), BGR->RGB through the table,
//      as {R,G,B,0} bf16 = 8 bytes per pixel, zero outside the frame (conv padding);
//   3. each wave computes 8 x 16 output pixels with v_mfma_f32_16x16x32_bf16: K = 9 taps x 4 channels
//      padded to 2 k-steps; the stem's whole weight matrix (2 KiB) stays in 2 fragment registers per lane
//      for the kernel's lifetime (weight-stationary), activations come from LDS as two 8-byte reads per
//      fragment (two taps x 4 channels);
//   4. bias + SiLU, 8-byte NHWC stores: the 4 lanes of a pixel write its 32 bytes contiguously.
#include "zly_internal.h"
#include "conv_device.h"
#include <stdlib.h>
#include <type_traits>

namespace zly {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define STEM_TH 16
#define STEM_TW 32
#define STEM_PH (STEM_TH * 2 + 1)
#define STEM_PW (STEM_TW * 2 + 1)

// A frame descriptor in ONE 16-byte scalar load.  Read field by field the compiler fetched src_off and w, compared w, and only then asked for h (the
// short-circuit of `w == tw && h == th`): a second dependent round trip to memory in front of every tile's first pixel load.
__device__ __forceinline__ FrameDesc load_desc(const FrameDesc* p)
{
    typedef unsigned int du32x4 __attribute__((ext_vector_type(4)));
    const du32x4 q = *reinterpret_cast<const du32x4*>(p);
    FrameDesc d;
    d.src_off = (unsigned long long)q[0] | ((unsigned long long)q[1] << 32);
    d.w = (int)q[2]; d.h = (int)q[3];
    return d;
}

// NT = output channel tiles of 16: 1 (YOLOv8n, 16-channel stem: rows in channel order) or 2 (YOLOv8-s, 32 channels: pair-permuted rows, a lane
// ends with 8 consecutive channels = one 16-byte store); the pixel fragments are read once for both tiles
template <int NT>
__global__ __launch_bounds__(256) void stem_fused_kernel(const StemArgs a)
{
    __shared__ __attribute__((aligned(16))) bf16x4 patch[STEM_PH * STEM_PW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kq = lane >> 4;
    const int f = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int oy0 = ty * STEM_TH, ox0 = tx * STEM_TW;
    const int iy0 = oy0 * 2 - 1, ix0 = ox0 * 2 - 1;

    // weight fragments: tiled [NT][2][lane][8] (k = tap*4 + c), resident in registers
    bf16x8 w[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) w[t][s] = *reinterpret_cast<const bf16x8*>(static_cast<const bf16_t*>(a.wgt) + (t * 2 + s) * 512 + lane * 8);

    const FrameDesc d = load_desc(&a.desc[f]);
    const float scale_w = (float)d.w / (float)a.tw;
    const float scale_h = (float)d.h / (float)a.th;
    const uint8_t* src = a.src + d.src_off;
    const bool same = (d.w == a.tw) & (d.h == a.th);
    const size_t frame_bytes = (size_t)d.w * d.h * 3;
    for (int u = tid; u < STEM_PH * STEM_PW; u += 256) {
        const int py = u / STEM_PW, px = u - py * STEM_PW;
        const int iy = iy0 + py, ix = ix0 + px;
        bf16x4 v = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        if ((unsigned)iy < (unsigned)a.th && (unsigned)ix < (unsigned)a.tw) {
            int sy = iy, sx = ix;
            if (!same) {       // request size == model size: the nearest-neighbour map is the identity (wave-uniform branch)
                sy = (int)((float)iy * scale_h); if (sy > d.h - 1) sy = d.h - 1;
                sx = (int)((float)ix * scale_w); if (sx > d.w - 1) sx = d.w - 1;
            }
            const size_t off = ((size_t)sy * d.w + sx) * 3;
            const uint8_t* q = src + off;
            // one (unaligned) 4-byte load instead of three byte loads -- this kernel is bound by instruction issue; the very
            // last pixel of a frame would read one byte past it and keeps the byte loads
            unsigned int px4;
            if (off + 4 <= frame_bytes) __builtin_memcpy(&px4, q, 4);                        // B | G<<8 | R<<16 | next B<<24 (amdhsa: unaligned global access is enabled)
            else px4 = (unsigned int)q[0] | ((unsigned int)q[1] << 8) | ((unsigned int)q[2] << 16);
            // BGR -> RGB; bf16(u8 * (1/255.f)) == bf16(u8 / 255.f) for all 256 values (tests/test_model_spec.py), so the
            // reference's divide (:693) + the bf16 rounding is one v_cvt_f32_ubyte + v_mul + convert, no table
            const float k = 1.0f / 255.0f;
            v[0] = (bf16_t)((float)((px4 >> 16) & 0xffu) * k); v[1] = (bf16_t)((float)((px4 >> 8) & 0xffu) * k); v[2] = (bf16_t)((float)(px4 & 0xffu) * k);
        }
        patch[u] = v;
    }
    __syncthreads();

    // taps of this lane: k-step s, fragment half j -> tap = s*8 + kq*2 + j (taps >= 9 have zero weights:
    // read any valid pixel)
    int toff[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tap = s * 8 + kq * 2 + j;
            const int ky = tap < 9 ? tap / 3 : 0, kx = tap < 9 ? tap - (tap / 3) * 3 : 0;
            toff[s][j] = ky * STEM_PW + kx;
        }

    f32x4 bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias[t] = *reinterpret_cast<const f32x4*>(a.bias + (NT == 1 ? kq * 4 : kq * 8 + t * 4));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 4 + (i >> 1), col = (i & 1) * 16 + p;
        const int base = (row * 2) * STEM_PW + col * 2;
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x4 lo = patch[base + toff[s][0]];
            const bf16x4 hi = patch[base + toff[s][1]];
            bf16x8 af;
            af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
            af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][s], af, acc[t], 0, 0, 0);
        }
        const int oy = oy0 + row, ox = ox0 + col;
        if (oy < a.Ho && ox < a.Wo) {
            const int m = (f * a.Ho + oy) * a.Wo + ox;
            bf16_t* dst = static_cast<bf16_t*>(a.out) + (m * a.out_cs + a.out_co);
            bf16x4 o[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 v = acc[t] + bias[t];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[r] * -1.442695041f));
                o[t] = __builtin_convertvector(v, bf16x4);
            }
            if (NT == 1) *reinterpret_cast<bf16x4*>(dst + kq * 4) = o[0];
            else {
                bf16x8 o8;
                o8[0] = o[0][0]; o8[1] = o[0][1]; o8[2] = o[0][2]; o8[3] = o[0][3];
                o8[4] = o[NT - 1][0]; o8[5] = o[NT - 1][1]; o8[6] = o[NT - 1][2]; o8[7] = o[NT - 1][3];
                *reinterpret_cast<bf16x8*>(dst + kq * 8) = o8;
            }
        }
    }
}

hipError_t launch_stem_fused(const StemArgs& a, int n, hipStream_t s)
{
    if (a.Cout != 16 && a.Cout != 32) return hipErrorInvalidValue;         // one or two 16-channel MFMA tiles (YOLOv8n / YOLOv8-s); wider stems use the generic path
    if (a.Cout == 32 && (a.out_cs % 8 || a.out_co % 8)) return hipErrorInvalidValue;
    const int tiles_y = (a.Ho + STEM_TH - 1) / STEM_TH;
    if (a.Cout == 16) hipLaunchKernelGGL(stem_fused_kernel<1>, dim3(a.tiles_x * tiles_y, n), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(stem_fused_kernel<2>, dim3(a.tiles_x * tiles_y, n), dim3(256), 0, s, a);
    return hipGetLastError();
}

int stem_tiles_x(int Wo) { return (Wo + STEM_TW - 1) / STEM_TW; }

// ------------------------------------------------------------------------------------------------
// preprocess + model.0 + model.1 as ONE kernel (bf16, 16-channel stem, 32-channel model.1: YOLOv8n).
//
// Un-fused, the stem's output (208 x 208 x 16 bf16 = 1.38 MB per frame at 416 x 416) is written to HBM and read back by
// model.1 (3x3 s2, 16 -> 32): 177 MB per 64-frame step, more than the request frames themselves (33 MB).  Here a
// workgroup owns a TH x TW tile of model.1 OUTPUTS of one frame and keeps the stem map it needs in LDS:
//   1. input patch (4TH+3) x (4TW+3), the reference's nearest-neighbour map + BGR->RGB + /255 (onnx_engine.cpp:673-693) as
//      {R,G,B,0} bf16, zero outside the model-sized image (the stem's padding)                       -> LDS
//   2. stem conv (3x3 s2, K = 27 -> 2 x 16x16x32 MFMA) over the (2TH+1) x (2TW+1) stem pixels the tile's 3x3 s2 windows
//      touch, bias + SiLU, bf16, ZERO outside the stem map (model.1's padding)                       -> LDS (pixel pitch 40 B:
//      conflict-free for the stride-2 8-byte fragment reads)     [same values the two-kernel path writes to HBM]
//   3. model.1 from the LDS map: one 16x16x16 MFMA per tap and channel tile (weights 9 KB in LDS), bias + SiLU -> HBM, one
//      16-byte store per lane (8 consecutive channels, pair-permuted rows as in the conv kernels).
// Pixels of both regions are linearised and cut into 16-pixel MFMA tiles (as bottleneck_pair_kernel does), so TW need not
// be a multiple of 16.  Stem pixels on tile borders are computed by two workgroups (8 % at 8 x 26 tiles).
// With a.dump (debug taps) every workgroup also writes the stem pixels it owns to the model.0 tensor.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4s;
#define STEM1_PITCH 40
#define STEM1_MAXIT 8      // input pixels staged per thread (8 waves: patches up to 4096 pixels)

// a value the compiler must recompute where it stands: per-lane index arithmetic of a rarely taken path is loop-invariant across a persistent workgroup's
// tiles, and hoisted out of the tile loop it holds registers (spills) in every phase
__device__ __forceinline__ int pin_here(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int div_small_s(int q, float inv) { return (int)(((float)q + 0.5f) * inv); }   // exact for q < 2^20, divisor < 2^10

#ifdef ZLY_STEM_DIAG
__device__ unsigned long long* g_stem_diag = nullptr;            // diagnostic build only (tools/stem_bench.hip): per-wave cycle sums of the phases
#define STEMSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); dsum[k] += t_ - dT0; dT0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STEMSTAMP(k) do { } while (0)
#endif

// Tap order of the stem weights for VAR >= 1 (STEM1_TAP_SLOT[ky * 3 + kx] = k slot of 4 channels; k = slot * 4 + c, two k-steps of 32): the 8-byte
// fragment halves of one ds_read_b64 -- lane groups kq 0/1 share the lanes 0-31, kq 2/3 the lanes 32-63 -- then hit disjoint banks.  A patch pixel is
// 2 dwords and the 16 pixels of an MFMA tile are 2 pixels apart, so one lane group covers the banks = 0, 1 (mod 4) of its tap's offset; its partner must
// read a tap an ODD number of pixels away (row pitch even: the parity of kx decides) or the very same addresses (zero-weight slots: broadcast).
//   read 0 (k-step 0, first half):  kq0 (0,0)  kq1 (0,1)  kq2 (1,0)  kq3 (1,1)
//   read 1 (k-step 0, second half): kq0 (2,0)  kq1 (2,1)  kq2 (0,2)  kq3 zero weights, reads kq2's pixel
//   read 2 (k-step 1, first half):  kq0 (1,2)  kq1 zero   kq2 (2,2)  kq3 zero            (k-step 1, second half: all zero weights, not read)
// Round 3's order (tap = s * 8 + kq * 2 + j) put two taps of equal pixel parity into half of the lane groups' pairs: 2-way conflicts on 4 of 8 half-wave
// reads, and a fourth read for slots whose weights are all zero.
static const int STEM1_TAP_SLOT[9] = {0, 2, 5, 4, 6, 8, 1, 3, 12};
const int* stem1_tap_slot() { return STEM1_TAP_SLOT; }

template <int NW, int VAR>
__global__ __launch_bounds__(NW * 64, NW == 16 ? 8 : NW == 12 ? 6 : 4) void stem_model1_kernel(const Stem1Args a)
{
    constexpr bool NEWP = VAR >= 1, PERS = VAR >= 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem1[];
    const int RH = 2 * a.TH + 1, RW = 2 * a.TW + 1;          // stem pixels the tile needs
    const int PH = 2 * RH + 1, PWV = 2 * RW + 1;             // input pixels those need
    // patch row pitch in pixels.  VAR >= 1: 4 TW + 6 = 2 (mod 4): rows are 16-byte aligned and consecutive rows are 16 bytes apart mod 32, which is what
    // makes the staging stores below conflict-free; it is even, which the tap order above relies on
    const int PW = NEWP ? PWV + 3 : PWV;
    unsigned char* lw = smem1;                               // model.1 weights: [2 tiles][9 taps][64 lanes][8 B]
    bf16x4* patch = reinterpret_cast<bf16x4*>(smem1 + 2 * 9 * 512);
    unsigned char* smap = reinterpret_cast<unsigned char*>(patch) + ((size_t)PH * PW * 8 + 15) / 16 * 16;
#ifdef ZLY_STEM_DIAG
    unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dT0 = __builtin_amdgcn_s_memtime();
    const unsigned long long dstart = dT0;
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kq = lane >> 4;
    // VAR <= 1: one tile per workgroup, grid (tiles of a frame, frames).  VAR 2 (PERS): grid (workgroups that fit the chip), a workgroup walks the tiles
    // b, b + gridDim.x, ... of the batch: weights / biases are fetched once per workgroup instead of once per tile, and the input bytes of the NEXT tile
    // are requested before this tile's convolutions, so that their memory round trip is no longer part of the staging phase.
    const int tpf = a.tiles_x * a.tiles_y;
    const int total = tpf * a.n;
    auto tile_origin = [&](int tile, int& f, int& oy1, int& ox1) {
        int b;
        if (PERS) { f = tile / tpf; b = tile - f * tpf; } else { f = blockIdx.y; b = blockIdx.x; }
        const int ty = b / a.tiles_x, tx = b - ty * a.tiles_x;
        oy1 = ty * a.TH; ox1 = tx * a.TW;                    // model.1 output tile origin
    };

    // model.1's weights (9 KB -> LDS, read in phase 3): REQUESTED here, stored after the first tile's pixels have been requested too (lw_store below).  As a
    // copy loop at this place it was load -> s_waitcnt vmcnt(0) -> ds_write per iteration: one or two exposed global round trips at the head of every one of
    // the 3328 workgroups of a batch-64 launch, before the frame descriptor and the pixels were even asked for.
    constexpr int LW_UNITS = 2 * 9 * 512 / 16, LW_T = (LW_UNITS + NW * 64 - 1) / (NW * 64);
    u32x4s lw_r[LW_T];
#pragma unroll
    for (int i = 0; i < LW_T; ++i) {
        const int u = min(tid + i * NW * 64, LW_UNITS - 1);                 // clamped: no branch around the load (a surplus thread re-reads the last unit)
        lw_r[i] = *reinterpret_cast<const u32x4s*>(static_cast<const unsigned char*>(a.w1) + (size_t)u * 16);
    }
    auto lw_store = [&]() {
#pragma unroll
        for (int i = 0; i < LW_T; ++i) {
            const int u = tid + i * NW * 64;
            if (u < LW_UNITS) *reinterpret_cast<u32x4s*>(lw + (size_t)u * 16) = lw_r[i];
        }
    };

    const bf16_t* wg0 = static_cast<const bf16_t*>(NEWP ? a.wgt0p : a.st.wgt);
    const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wg0 + lane * 8);
    const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wg0 + 512 + lane * 8);

    // VAR >= 1: the reciprocals come from the host (IEEE divides there give the same bits; six divides per thread were ~70 of a thread's ~630 VALU instructions)
    const float invPW = NEWP ? a.inv_pw : 1.0f / (float)PWV, invRW = NEWP ? a.inv_rw : 1.0f / (float)RW, invTW = NEWP ? a.inv_tw : 1.0f / (float)a.TW;

    // Quad staging (request size == model size, the metric's configuration; VAR >= 1): the resize map is the identity and a patch row is PWV * 3
    // contiguous bytes of the frame.  A thread takes FOUR consecutive pixels (a quad), twice: one 12-byte load, v_cvt_f32_ubyte0..3 straight off the
    // dwords, 12 multiplies, packed converts, TWO 16-byte LDS stores.  Lanes -> quads so that the stores are conflict-free (ds_write_b128 is served in
    // groups of 8 consecutive lanes over 32 banks = eight 16-byte slots): the 8 lanes of a group take quads 4 qb .. 4 qb + 3 of patch rows 2 rp
    // (lanes 0-3) and 2 rp + 1 (lanes 4-7).  A quad is 32 bytes, so one row's four first halves fill slots 0, 2, 4, 6; the next row starts 16 bytes
    // later mod 32 (row pitch = 2 mod 4 pixels) and fills 1, 3, 5, 7.  (Round 3: consecutive lanes took consecutive quads with four 8-byte stores
    // each, whose 16-lane groups hit 8 of 32 banks: 4-way conflicts, 62 % of this kernel's SQ_LDS_BANK_CONFLICT.)
    // The lane -> quad map does not depend on the tile (the host checks that two quads per thread cover the patch): computed once.
    // a tile's input bytes in flight: two quads per thread, plain scalars (an indexed struct was left in scratch memory by the compiler, with a wait for the
    // loads in front of the scratch store); mode 0: nothing to do, 1: fast quad, 2: per-pixel quad (frame border)
    int qpy[2] = {0, 0}, qpx[2] = {0, 0};
    bool qok[2] = {false, false};
    if (NEWP) {
        const int QW = (PWV + 3) >> 2;                           // quads per patch row that hold pixels
        const int QB = (QW + 3) >> 2;                            // blocks of four quads per row
        const int NG = ((PH + 1) >> 1) * QB;                     // 8-lane groups = (row pair, quad block)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = k * NW * 64 + tid;
            const int G = u >> 3;
            const int rp = div_small_s(G, a.inv_qb);
            qpy[k] = 2 * rp + ((u >> 2) & 1); qpx[k] = (4 * (G - rp * QB) + (u & 3)) * 4;
            qok[k] = G < NG && qpy[k] < PH && qpx[k] < PWV;
        }
    }
    unsigned int pa0 = 0, pa1 = 0, pa2 = 0, pb0 = 0, pb1 = 0, pb2 = 0;
    int pma = 0, pmb = 0;
    // One quad: 12 bytes from (iy, ix clamped into the row), whatever the quad's position.  Returns 0: no quad, 3: row outside the image (zeros),
    // 1: the window IS the quad, 16 + (ix - clamped ix + 3): the quad hangs over the left / right frame border by 1-3 pixels -- the window is then shifted
    // by whole pixels at conversion time (zero bytes move in = the zero padding).  Round 4 up to here took such quads pixel by pixel with one byte-wise
    // global round trip each: every row of the half of all tiles that touch the left or right border had one.
    auto issue_quad = [&](bool ok, int py, int px, int iy0, int ix0, const FrameDesc& d, const uint8_t* src, unsigned int& r0, unsigned int& r1, unsigned int& r2) -> int {
        if (!ok) return 0;
        const int iy = iy0 + py, ix = ix0 + px;
        if ((unsigned)iy >= (unsigned)a.st.th || ix >= a.st.tw || ix + 3 < 0) return 3;
        const int ixc = min(max(ix, 0), a.st.tw - 4);
        const uint8_t* q = src + ((size_t)iy * d.w + ixc) * 3;
        typedef unsigned int u32x3 __attribute__((ext_vector_type(3), aligned(1)));
        const u32x3 v = *reinterpret_cast<const u32x3*>(q);          // 12 bytes: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3 (unaligned global access is enabled on amdhsa)
        r0 = v[0]; r1 = v[1]; r2 = v[2];
        return ix == ixc ? 1 : 16 + (ix - ixc + 3);
    };
    auto issue_quads = [&](int tile) {
        pma = 0; pmb = 0;
        if (!NEWP) return;
        int f, oy1, ox1;
        tile_origin(tile, f, oy1, ox1);
        const int iy0 = 4 * oy1 - 3, ix0 = 4 * ox1 - 3;
        const FrameDesc d = load_desc(&a.st.desc[f]);
        if (!((d.w == a.st.tw) & (d.h == a.st.th))) return;     // resized frame: the general path below does its own loads
        const uint8_t* src = a.st.src + d.src_off;
        pma = issue_quad(qok[0], qpy[0], qpx[0], iy0, ix0, d, src, pa0, pa1, pa2);
        pmb = issue_quad(qok[1], qpy[1], qpx[1], iy0, ix0, d, src, pb0, pb1, pb2);
    };

    const f32x4 bias0 = *reinterpret_cast<const f32x4*>(a.st.bias + kq * 4);
    const f32x4 b1lo = *reinterpret_cast<const f32x4*>(a.b1 + kq * 8), b1hi = *reinterpret_cast<const f32x4*>(a.b1 + kq * 8 + 4);
    // model.1's weights of both channel tiles for the 9 taps: registers for the kernel's last phase (36 VGPRs) instead of 18 LDS reads per pixel tile
    constexpr bool WREG = NW <= 8;                           // more waves per workgroup: the register budget goes to occupancy instead
    const unsigned char* wl = lw + lane * 8;
    s16x4 wa[WREG ? 9 : 1], wb[WREG ? 9 : 1];
    auto load_wregs = [&]() {
        if (WREG) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                wa[k % (WREG ? 9 : 1)] = *reinterpret_cast<const s16x4*>(wl + (0 * 9 + k) * 512);
                wb[k % (WREG ? 9 : 1)] = *reinterpret_cast<const s16x4*>(wl + (1 * 9 + k) * 512);
            }
        }
    };

    int tile = PERS ? (int)blockIdx.x : (int)(blockIdx.y * tpf + blockIdx.x);
    issue_quads(tile);
    lw_store();                                              // (the two barriers of the first tile lie between this store and the first read)
    for (; tile < total; tile += PERS ? (int)gridDim.x : total) {
    int f, oy1, ox1;
    tile_origin(tile, f, oy1, ox1);
    const int sy0 = 2 * oy1 - 1, sx0 = 2 * ox1 - 1;          // stem region origin (stem output coordinates)
    const int iy0 = 2 * sy0 - 1, ix0 = 2 * sx0 - 1;          // input patch origin (model-input pixel coordinates)

    // ---- 1. input patch ------------------------------------------------------------------------------------------
    const FrameDesc d = load_desc(&a.st.desc[f]);
    const uint8_t* src = a.st.src + d.src_off;
    const bool same = (d.w == a.st.tw) & (d.h == a.st.th);
    const size_t frame_bytes = (size_t)d.w * d.h * 3;
    STEMSTAMP(0);
    if (same && NEWP) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int mode_k = k ? pmb : pma;
            if (mode_k == 0) continue;
            unsigned int x0 = k ? pb0 : pa0, x1 = k ? pb1 : pa1, x2 = k ? pb2 : pa2;
            if (mode_k == 3) { x0 = 0u; x1 = 0u; x2 = 0u; }
            if (mode_k >= 16) {
                const int sft = mode_k - 19;                     // pixels; > 0: the quad starts right of the window (right frame border), < 0: left of it
                const int kb = 3 * (sft < 0 ? -sft : sft), dw = kb >> 2, b = kb & 3;      // 3 / 6 / 9 bytes: b = 3 / 2 / 1
                if (sft > 0) {                                   // window bytes move down by kb, zeros come in on top
                    const unsigned int z0 = dw == 0 ? x0 : dw == 1 ? x1 : x2, z1 = dw == 0 ? x1 : dw == 1 ? x2 : 0u, z2 = dw == 0 ? x2 : 0u;
                    x0 = __builtin_amdgcn_alignbyte(z1, z0, b); x1 = __builtin_amdgcn_alignbyte(z2, z1, b); x2 = __builtin_amdgcn_alignbyte(0u, z2, b);
                } else {                                         // up by kb, zeros come in below
                    const unsigned int w2 = dw == 0 ? x2 : dw == 1 ? x1 : x0, w1 = dw == 0 ? x1 : dw == 1 ? x0 : 0u, w0 = dw == 0 ? x0 : 0u;
                    x2 = __builtin_amdgcn_alignbyte(w2, w1, 4 - b); x1 = __builtin_amdgcn_alignbyte(w1, w0, 4 - b); x0 = __builtin_amdgcn_alignbyte(w0, 0u, 4 - b);
                }
            }
            {
                const float kk = 1.0f / 255.0f;              // bf16(u8 * (1/255.f)) == bf16(u8 / 255.f) for all 256 values (tests/test_model_spec.py)
                const float b0 = (float)(x0 & 0xffu) * kk, g0 = (float)((x0 >> 8) & 0xffu) * kk, rr0 = (float)((x0 >> 16) & 0xffu) * kk;
                const float b1 = (float)(x0 >> 24) * kk, g1 = (float)(x1 & 0xffu) * kk, rr1 = (float)((x1 >> 8) & 0xffu) * kk;
                const float b2 = (float)((x1 >> 16) & 0xffu) * kk, g2 = (float)(x1 >> 24) * kk, rr2 = (float)(x2 & 0xffu) * kk;
                const float b3 = (float)((x2 >> 8) & 0xffu) * kk, g3 = (float)((x2 >> 16) & 0xffu) * kk, rr3 = (float)(x2 >> 24) * kk;
                bf16x8 lo, hi;
                lo[0] = (bf16_t)rr0; lo[1] = (bf16_t)g0; lo[2] = (bf16_t)b0; lo[3] = (bf16_t)0.f; lo[4] = (bf16_t)rr1; lo[5] = (bf16_t)g1; lo[6] = (bf16_t)b1; lo[7] = (bf16_t)0.f;
                hi[0] = (bf16_t)rr2; hi[1] = (bf16_t)g2; hi[2] = (bf16_t)b2; hi[3] = (bf16_t)0.f; hi[4] = (bf16_t)rr3; hi[5] = (bf16_t)g3; hi[6] = (bf16_t)b3; hi[7] = (bf16_t)0.f;
                bf16x8* dst = reinterpret_cast<bf16x8*>(patch + qpy[k] * PW + qpx[k]);      // 32-byte aligned: rows are 16-byte aligned (PW even), quads 32 bytes
                dst[0] = lo;
                dst[1] = hi;
            }
        }
    } else if (same) {
        // Request size == model size (the metric's configuration): the resize map is the identity and a patch row is PW * 3 contiguous
        // bytes of the frame.  A thread takes FOUR consecutive pixels: one 12-byte load, v_cvt_f32_ubyte0..3 straight off the dwords, 12
        // multiplies, packed converts, two 16-byte LDS stores -- ~9 VALU per pixel.  (The general path below spends ~40 per pixel on the
        // per-pixel index map, bounds tests, 64-bit addressing and byte shuffles; this kernel is VALU-issue bound.)  Quads that touch the
        // frame border (or the patch's last, partial quad) take the per-pixel path.
        const int QW = (PW + 3) >> 2;                            // quads per patch row
        const float invQW = 1.0f / (float)QW;
        for (int u0 = 0; u0 < PH * QW; u0 += NW * 64 * 2) {
            unsigned int r0[2][3];
            int mode[2], pyq[2], pxq[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int u = u0 + k * NW * 64 + tid;
                mode[k] = 0;                                     // 0: nothing to do, 1: fast quad, 2: per-pixel quad
                if (u < PH * QW) {
                    const int py = div_small_s(u, invQW), q4 = (u - py * QW) * 4;
                    pyq[k] = py; pxq[k] = q4;
                    const int iy = iy0 + py, ix = ix0 + q4;
                    const bool row_in = (unsigned)iy < (unsigned)a.st.th;
                    const bool fast = row_in && ix >= 0 && ix + 3 < a.st.tw && q4 + 3 < PW && ((size_t)iy * d.w + ix) * 3 + 12 <= frame_bytes;
                    mode[k] = fast ? 1 : 2;
                    if (fast) {
                        const uint8_t* q = src + ((size_t)iy * d.w + ix) * 3;
                        typedef unsigned int u32x3 __attribute__((ext_vector_type(3), aligned(1)));
                        const u32x3 v = *reinterpret_cast<const u32x3*>(q);          // 12 bytes: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3 (unaligned global access is enabled on amdhsa)
                        r0[k][0] = v[0]; r0[k][1] = v[1]; r0[k][2] = v[2];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (mode[k] == 1) {
                    const float kk = 1.0f / 255.0f;              // bf16(u8 * (1/255.f)) == bf16(u8 / 255.f) for all 256 values (tests/test_model_spec.py)
                    const unsigned int w0 = r0[k][0], w1 = r0[k][1], w2 = r0[k][2];
                    const float b0 = (float)(w0 & 0xffu) * kk, g0 = (float)((w0 >> 8) & 0xffu) * kk, rr0 = (float)((w0 >> 16) & 0xffu) * kk;
                    const float b1 = (float)(w0 >> 24) * kk, g1 = (float)(w1 & 0xffu) * kk, rr1 = (float)((w1 >> 8) & 0xffu) * kk;
                    const float b2 = (float)((w1 >> 16) & 0xffu) * kk, g2 = (float)(w1 >> 24) * kk, rr2 = (float)(w2 & 0xffu) * kk;
                    const float b3 = (float)((w2 >> 8) & 0xffu) * kk, g3 = (float)((w2 >> 16) & 0xffu) * kk, rr3 = (float)(w2 >> 24) * kk;
                    bf16x8 lo, hi;
                    lo[0] = (bf16_t)rr0; lo[1] = (bf16_t)g0; lo[2] = (bf16_t)b0; lo[3] = (bf16_t)0.f; lo[4] = (bf16_t)rr1; lo[5] = (bf16_t)g1; lo[6] = (bf16_t)b1; lo[7] = (bf16_t)0.f;
                    hi[0] = (bf16_t)rr2; hi[1] = (bf16_t)g2; hi[2] = (bf16_t)b2; hi[3] = (bf16_t)0.f; hi[4] = (bf16_t)rr3; hi[5] = (bf16_t)g3; hi[6] = (bf16_t)b3; hi[7] = (bf16_t)0.f;
                    bf16x4* dst = patch + pyq[k] * PW + pxq[k];      // 8 bytes per pixel: pairs are 16-byte stores only when aligned -- use four-pixel-safe 8-byte pairs
                    reinterpret_cast<bf16x4*>(dst)[0] = bf16x4{lo[0], lo[1], lo[2], lo[3]};
                    reinterpret_cast<bf16x4*>(dst)[1] = bf16x4{lo[4], lo[5], lo[6], lo[7]};
                    reinterpret_cast<bf16x4*>(dst)[2] = bf16x4{hi[0], hi[1], hi[2], hi[3]};
                    reinterpret_cast<bf16x4*>(dst)[3] = bf16x4{hi[4], hi[5], hi[6], hi[7]};
                } else if (mode[k] == 2) {
                    for (int j = 0; j < 4; ++j) {
                        const int pxx = pxq[k] + j;
                        if (pxx >= PW) break;
                        const int iy = iy0 + pyq[k], ix = ix0 + pxx;
                        bf16x4 v = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                        if ((unsigned)iy < (unsigned)a.st.th && (unsigned)ix < (unsigned)a.st.tw) {
                            const uint8_t* q = src + ((size_t)iy * d.w + ix) * 3;
                            const float kk = 1.0f / 255.0f;
                            v[0] = (bf16_t)((float)q[2] * kk); v[1] = (bf16_t)((float)q[1] * kk); v[2] = (bf16_t)((float)q[0] * kk);
                        }
                        patch[pyq[k] * PW + pxx] = v;
                    }
                }
            }
        }
    } else {
        // all of a thread's loads are issued before the first conversion (the loop below is fully unrolled: STEM1_MAXIT pixels per thread):
        // with one load per loop iteration every iteration exposed a full memory round trip, ~6 us of the ~12 us a tile took
        const float scale_w = (float)d.w / (float)a.st.tw;
        const float scale_h = (float)d.h / (float)a.st.th;
        unsigned int raw[STEM1_MAXIT];
        const int tidg = PERS ? pin_here(tid) : tid;
    #pragma unroll
        for (int k = 0; k < STEM1_MAXIT; ++k) {
            const int u = tidg + k * NW * 64;
            raw[k] = 0x80000000u;                                   // bit 31: pixel outside the model-sized image (or beyond the patch) -> zeros
            if (u < PH * PWV) {
                const int py = div_small_s(u, invPW), px = u - py * PWV;
                const int iy = iy0 + py, ix = ix0 + px;
                if ((unsigned)iy < (unsigned)a.st.th && (unsigned)ix < (unsigned)a.st.tw) {
                    int sy = iy, sx = ix;
                    if (!same) {
                        sy = (int)((float)iy * scale_h); if (sy > d.h - 1) sy = d.h - 1;
                        sx = (int)((float)ix * scale_w); if (sx > d.w - 1) sx = d.w - 1;
                    }
                    const size_t off = ((size_t)sy * d.w + sx) * 3;
                    const uint8_t* q = src + off;
                    unsigned int px4;
                    if (off + 4 <= frame_bytes) __builtin_memcpy(&px4, q, 4);        // B | G<<8 | R<<16 | next B<<24 (unaligned global access is enabled on amdhsa)
                    else px4 = (unsigned int)q[0] | ((unsigned int)q[1] << 8) | ((unsigned int)q[2] << 16);
                    raw[k] = px4 & 0x00ffffffu;
                }
            }
        }
    #pragma unroll
        for (int k = 0; k < STEM1_MAXIT; ++k) {
            const int u = tidg + k * NW * 64;
            if (u < PH * PWV) {
                bf16x4 v = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                if (!(raw[k] & 0x80000000u)) {
                    const float kk = 1.0f / 255.0f;                  // bf16(u8 * (1/255.f)) == bf16(u8 / 255.f) for all 256 values (tests/test_model_spec.py)
                    v[0] = (bf16_t)((float)((raw[k] >> 16) & 0xffu) * kk); v[1] = (bf16_t)((float)((raw[k] >> 8) & 0xffu) * kk); v[2] = (bf16_t)((float)(raw[k] & 0xffu) * kk);
                }
                const int py = div_small_s(u, invPW);
                patch[u + py * (PW - PWV)] = v;
            }
        }
    }
    if (PERS && tile + (int)gridDim.x < total) issue_quads(tile + (int)gridDim.x);      // in flight across both convolutions of this tile
    STEMSTAMP(1);
    __syncthreads();
    STEMSTAMP(2);

    // ---- 2. stem conv over the region -> LDS map ---------------------------------------------------------------------
    int toff[2][2];
    if (NEWP) {
        // STEM1_TAP_SLOT's order (above): reads 0 / 1 are the halves of k-step 0, read 2 the first half of k-step 1; k-step 1's second half is zero
        const int ky0 = kq >> 1, kx0 = kq & 1;                                                          // (0,0) (0,1) (1,0) (1,1)
        const int ky1 = kq < 2 ? 2 : 0, kx1 = kq < 2 ? kq : 2;                                          // (2,0) (2,1) (0,2) (0,2)
        const int ky2 = kq < 2 ? 1 : 2;                                                                 // (1,2) (1,2) (2,2) (2,2)
        toff[0][0] = ky0 * PW + kx0; toff[0][1] = ky1 * PW + kx1; toff[1][0] = ky2 * PW + 2; toff[1][1] = 0;
    } else {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tap = s * 8 + kq * 2 + j;                 // taps >= 9 have zero weights: read any valid pixel
                const int ky = tap < 9 ? tap / 3 : 0, kx = tap < 9 ? tap - (tap / 3) * 3 : 0;
                toff[s][j] = ky * PW + kx;
            }
    }
    const int NR = RH * RW, ntR = (NR + 15) >> 4;
    // One 16-pixel tile of the region: fragments -> 2 MFMAs -> bias + SiLU -> LDS map.  Lanes beyond the region's last pixel work on a copy of
    // it (same inputs, same result, same address: the store needs no guard), so the body has no branch and two tiles can be interleaved:
    // every step of a tile (LDS read -> MFMA -> exp -> rcp -> store) waits for the one before, and with one tile per iteration a wave had
    // nothing to issue meanwhile.
    auto stem_frags = [&](int t, bf16x8 (&af)[2], int& qc, int& ry, int& rx) {
        qc = min(t * 16 + p, NR - 1);
        ry = div_small_s(qc, invRW); rx = qc - __mul24(ry, RW);           // 24-bit multiplies: full rate (v_mul_lo_u32 is a quarter-rate instruction)
        const int base = __mul24(2 * ry, PW) + 2 * rx;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x4 lo = patch[base + toff[s][0]];
            bf16x4 hi = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            if (!(NEWP && s == 1)) hi = patch[base + toff[s][1]];
            af[s][0] = lo[0]; af[s][1] = lo[1]; af[s][2] = lo[2]; af[s][3] = lo[3];
            af[s][4] = hi[0]; af[s][5] = hi[1]; af[s][6] = hi[2]; af[s][7] = hi[3];
        }
    };
    // epilogue of a tile.  BORDER: the region reaches beyond the stem map (workgroups of the first tile row / column): those pixels are the
    // zero padding of model.1, not conv values; the other 2/3 of the workgroups skip the test and the select.  Conversions are packed (two
    // values per v_cvt_pk_bf16_f32) and the select acts on the packed dwords: 62 -> ~47 VALU instructions per tile of a kernel that is
    // bound by VALU issue.
    const bool dump = a.dump != 0;
    auto stem_finish = [&](auto border_tag, f32x4 acc, int qc, int ry, int rx) {
        constexpr bool BORDER = decltype(border_tag)::value;
        f32x4 v = acc + bias0;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[r] * -1.442695041f));
        bf16x4 o = __builtin_convertvector(v, bf16x4);
        const int sy = sy0 + ry, sx = sx0 + rx;
        bool inmap = true;
        if (BORDER) {
            inmap = (unsigned)sy < (unsigned)a.st.Ho && (unsigned)sx < (unsigned)a.st.Wo;
            const bf16x4 z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            o = inmap ? o : z;
        }
        *reinterpret_cast<bf16x4*>(smap + __mul24(qc, STEM1_PITCH) + kq * 8) = o;
        // debug taps: the stem pixels this tile owns (rows / columns 1 .. 2TH / 2TW of the region) also go to the model.0 tensor
        if (dump) {
            if (inmap && ry >= 1 && rx >= 1)
                *reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(a.st.out) + (((size_t)f * a.st.Ho + sy) * a.st.Wo + sx) * a.st.out_cs + a.st.out_co + kq * 4) = o;
        }
    };
    auto phase2 = [&](auto border_tag) {
        int t = wave;
        for (; t + NW < ntR; t += 2 * NW) {
            bf16x8 af0[2], af1[2];
            int q0, ry0, rx0, q1, ry1, rx1;
            stem_frags(t, af0, q0, ry0, rx0);
            stem_frags(t + NW, af1, q1, ry1, rx1);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, af0[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, af1[0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, af0[1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, af1[1], acc1, 0, 0, 0);
            stem_finish(border_tag, acc0, q0, ry0, rx0);
            stem_finish(border_tag, acc1, q1, ry1, rx1);
        }
        if (t < ntR) {
            bf16x8 af0[2];
            int q0, ry0, rx0;
            stem_frags(t, af0, q0, ry0, rx0);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, af0[0], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, af0[1], acc0, 0, 0, 0);
            stem_finish(border_tag, acc0, q0, ry0, rx0);
        }
    };
    if (sy0 >= 0 && sx0 >= 0 && sy0 + RH <= a.st.Ho && sx0 + RW <= a.st.Wo) phase2(std::false_type{});
    else phase2(std::true_type{});
    STEMSTAMP(3);
    __syncthreads();
    STEMSTAMP(4);

    // ---- 3. model.1 (3x3 s2, 16 -> 32) from the LDS map --------------------------------------------------------------------
    const int NO = a.TH * a.TW, ntO = (NO + 15) >> 4;
    load_wregs();                                            // per tile (LDS reads): held across the loop they cost 36 VGPRs of every phase's budget
    const int kxo[3] = {0, opaque_offset(STEM1_PITCH), opaque_offset(2 * STEM1_PITCH)};
    auto m1_tile = [&](int t, f32x4& acc0, f32x4& acc1, int& oy, int& ox) {
        const int qc = min(t * 16 + p, NO - 1);
        oy = div_small_s(qc, invTW); ox = qc - __mul24(oy, a.TW);
        const unsigned char* row0 = smap + __mul24(__mul24(2 * oy, RW) + 2 * ox, STEM1_PITCH) + kq * 8;
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const s16x4 x = *reinterpret_cast<const s16x4*>(row0 + __mul24(ky, RW * STEM1_PITCH) + kxo[kx]);      // kx offsets opaque: never ds_read2_b64, which is 2-way conflicted at this pitch (conv_device.h)
                const s16x4 fa = WREG ? wa[(ky * 3 + kx) % (WREG ? 9 : 1)] : *reinterpret_cast<const s16x4*>(wl + (0 * 9 + ky * 3 + kx) * 512);
                const s16x4 fb = WREG ? wb[(ky * 3 + kx) % (WREG ? 9 : 1)] : *reinterpret_cast<const s16x4*>(wl + (1 * 9 + ky * 3 + kx) * 512);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fa, x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fb, x, acc1, 0, 0, 0);
            }
    };
    auto m1_finish = [&](int t, f32x4 acc0, f32x4 acc1, int oy, int ox) {
        const int q = t * 16 + p;
        const int gy = oy1 + oy, gx = ox1 + ox;
        if (q < NO && gy < a.H1 && gx < a.W1) {
            f32x4 lo = acc0 + b1lo, hi = acc1 + b1hi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lo[r] = lo[r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(lo[r] * -1.442695041f));
                hi[r] = hi[r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(hi[r] * -1.442695041f));
            }
            const bf16x8 o = __builtin_convertvector(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7), bf16x8);
            *reinterpret_cast<bf16x8*>(static_cast<bf16_t*>(a.out1) + (((size_t)f * a.H1 + gy) * a.W1 + gx) * a.out1_cs + a.out1_co + kq * 8) = o;
        }
    };
    // a wave's tiles two at a time (four independent MFMA chains, two epilogues to interleave), as in phase 2
    int t3 = wave;
    for (; t3 + NW < ntO; t3 += 2 * NW) {
        f32x4 a0, a1, c0, c1;
        int oyA, oxA, oyB, oxB;
        m1_tile(t3, a0, a1, oyA, oxA);
        m1_tile(t3 + NW, c0, c1, oyB, oxB);
        m1_finish(t3, a0, a1, oyA, oxA);
        m1_finish(t3 + NW, c0, c1, oyB, oxB);
    }
    if (t3 < ntO) {
        f32x4 a0, a1;
        int oyA, oxA;
        m1_tile(t3, a0, a1, oyA, oxA);
        m1_finish(t3, a0, a1, oyA, oxA);
    }
    STEMSTAMP(5);
    if (!PERS) break;
    }   // tiles of this workgroup
#ifdef ZLY_STEM_DIAG
    if (lane == 0 && g_stem_diag) {
        unsigned long long* o = g_stem_diag + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
        for (int i = 0; i < 6; ++i) o[i] = dsum[i];
        o[6] = __builtin_amdgcn_s_memtime() - dstart;
    }
#endif
}

static constexpr int STEM1_NW = 8;
static size_t stem1_lds_bytes(int th, int tw, int var)
{
    const size_t RH = 2 * th + 1, RW = 2 * tw + 1, PH = 2 * RH + 1, PW = 2 * RW + 1 + (var >= 1 ? 3 : 0);
    return 2 * 9 * 512 + (PH * PW * 8 + 15) / 16 * 16 + RH * RW * STEM1_PITCH;
}

// tile shape: the widest TW <= 26 that divides the map width if there is one (104 -> 26, 160 -> 20), TH = 8
void stem1_plan(int H1, int W1, int* th, int* tw)
{
    *th = 8;
    *tw = 26;
    for (int c = 26; c >= 13; --c)
        if (W1 % c == 0) { *tw = c; break; }
    if (const char* v = getenv("ZLY_STEM1_TW")) { if (atoi(v) >= 8 && atoi(v) <= 26) *tw = atoi(v); }      // tuning aids
    if (const char* v = getenv("ZLY_STEM1_TH")) { if (atoi(v) >= 2 && atoi(v) <= 8) *th = atoi(v); }
    (void)H1;
}

typedef void (*stem1_fn)(const Stem1Args);
static stem1_fn pick_stem1(int nw, int var)
{
    if (var >= 2) return nw == 12 ? stem_model1_kernel<12, 2> : nw == 16 ? stem_model1_kernel<16, 2> : stem_model1_kernel<STEM1_NW, 2>;
    if (var == 1) return nw == 12 ? stem_model1_kernel<12, 1> : nw == 16 ? stem_model1_kernel<16, 1> : stem_model1_kernel<STEM1_NW, 1>;
    return nw == 12 ? stem_model1_kernel<12, 0> : nw == 16 ? stem_model1_kernel<16, 0> : stem_model1_kernel<STEM1_NW, 0>;
}

static int g_stem1_cus = 256;
hipError_t stem1_init()
{
    for (int var = 0; var <= 2; ++var)
        for (int nw : {STEM1_NW, 12, 16}) {
            hipError_t r = hipFuncSetAttribute((const void*)pick_stem1(nw, var), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (r != hipSuccess) return r;
        }
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) g_stem1_cus = cus;
    return hipSuccess;
}

// a.nw: waves per workgroup (8; 12 / 16 = tuning aid ZLY_STEM1_NW); a.var: 2 = persistent workgroups with the next tile's input bytes in flight (default),
// 1 = one tile per workgroup, 0 = round 3's staging / tap order as well (ZLY_STEM1_VAR, A/B on one box).  Both are read by the engine once per
// zly_create, not here (a process-static switch cannot be toggled by a test)
hipError_t launch_stem_model1(const Stem1Args& a0, int n, hipStream_t s)
{
    Stem1Args a = a0;
    if (a.st.Cout != 16 || a.TH < 1 || a.TW < 1 || a.out1_cs % 8 || a.out1_co % 8) return hipErrorInvalidValue;
    if (a.H1 * 2 != a.st.Ho || a.W1 * 2 != a.st.Wo) return hipErrorInvalidValue;      // even stem map: model.1 output = half of it
    // small batches (the latency path): the planned 8 x 26 tiles are 52 workgroups per frame -- a fifth of the chip at batch 1, each walking 8 tile rounds per
    // phase.  Halve the tile (width first, then height) while the launch has fewer workgroups than the chip holds: batch 1 step 0.1857 -> 0.1832 ms at 8 x 13.
    // (a.small_tiles = 0: the engine's ZLY_STEM1_TW / ZLY_STEM1_TH were given, keep the planned shape.)
    if (a.small_tiles) {
        while ((long long)a.tiles_x * a.tiles_y * n < 2LL * g_stem1_cus) {
            if (a.TW > 13 && a.TW % 2 == 0) a.TW /= 2;
            else if (a.TH > 4 && a.TH % 2 == 0) a.TH /= 2;
            else break;
            a.tiles_x = (a.W1 + a.TW - 1) / a.TW; a.tiles_y = (a.H1 + a.TH - 1) / a.TH;
        }
    }
    const int nw = (a.nw == 12 || a.nw == 16) ? a.nw : STEM1_NW, var = a.var >= 2 ? 2 : a.var == 1 ? 1 : 0;
    if (var >= 1 && !a.wgt0p) return hipErrorInvalidValue;
    const size_t lds = stem1_lds_bytes(a.TH, a.TW, var);
    if (lds > 160 * 1024 || (4 * a.TH + 3) * (4 * a.TW + 3) > STEM1_MAXIT * STEM1_NW * 64) return hipErrorInvalidValue;
    const int ph = 4 * a.TH + 3, pwv = 4 * a.TW + 3, qb = ((pwv + 3) / 4 + 3) / 4;
    if (var >= 1 && ((ph + 1) / 2) * qb * 8 > nw * 64 * 2) return hipErrorInvalidValue;    // the quad staging takes two quads per thread
    a.inv_pw = 1.0f / (float)pwv; a.inv_rw = 1.0f / (float)(2 * a.TW + 1); a.inv_tw = 1.0f / (float)a.TW; a.inv_qb = 1.0f / (float)qb;
    a.n = n;
    const long long total = (long long)a.tiles_x * a.tiles_y * n;
    if (total > 0x7fffffff) return hipErrorInvalidValue;
    if (var >= 2) {
        // as many workgroups as stay resident (LDS: 160 KB per CU); with fewer tiles than that, one tile each
        long long wgs = (long long)g_stem1_cus * (long long)((160 * 1024) / lds < 1 ? 1 : (160 * 1024) / lds);
        if (a.pgrid > 0) wgs = a.pgrid;                                                            // tuning aid ZLY_STEM1_GRID (read by the engine per zly_create)
        if (wgs > total) wgs = total;
        hipLaunchKernelGGL(pick_stem1(nw, var), dim3((unsigned)wgs), dim3(nw * 64), lds, s, a);
    } else {
        hipLaunchKernelGGL(pick_stem1(nw, var), dim3(a.tiles_x * a.tiles_y, n), dim3(nw * 64), lds, s, a);
    }
    return hipGetLastError();
}

}  // namespace zly
