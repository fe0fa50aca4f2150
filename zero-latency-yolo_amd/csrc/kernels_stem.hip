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

namespace zly {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define STEM_TH 16
#define STEM_TW 32
#define STEM_PH (STEM_TH * 2 + 1)
#define STEM_PW (STEM_TW * 2 + 1)

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

    // weight fragments: tiled [1][2][lane][8] (k = tap*4 + c), resident in registers
    const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(static_cast<const bf16_t*>(a.wgt) + lane * 8);
    const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(static_cast<const bf16_t*>(a.wgt) + 512 + lane * 8);

    const FrameDesc d = a.desc[f];
    const float scale_w = (float)d.w / (float)a.tw;
    const float scale_h = (float)d.h / (float)a.th;
    const uint8_t* src = a.src + d.src_off;
    const bool same = d.w == a.tw && d.h == a.th;
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

    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + kq * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 4 + (i >> 1), col = (i & 1) * 16 + p;
        const int base = (row * 2) * STEM_PW + col * 2;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x4 lo = patch[base + toff[s][0]];
            const bf16x4 hi = patch[base + toff[s][1]];
            bf16x8 af;
            af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
            af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s == 0 ? w0 : w1, af, acc, 0, 0, 0);
        }
        const int oy = oy0 + row, ox = ox0 + col;
        if (oy < a.Ho && ox < a.Wo && kq * 4 < a.Cout) {
            f32x4 v = acc + bias;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[r] * -1.442695041f));
            bf16x4 o;
            o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
            const int m = (f * a.Ho + oy) * a.Wo + ox;
            *reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(a.out) + (m * a.out_cs + a.out_co + kq * 4)) = o;
        }
    }
}

hipError_t launch_stem_fused(const StemArgs& a, int n, hipStream_t s)
{
    if (a.Cout != 16) return hipErrorInvalidValue;         // one 16-channel MFMA tile (YOLOv8n); wider stems use the generic path
    const int tiles_y = (a.Ho + STEM_TH - 1) / STEM_TH;
    hipLaunchKernelGGL(stem_fused_kernel, dim3(a.tiles_x * tiles_y, n), dim3(256), 0, s, a);
    return hipGetLastError();
}

int stem_tiles_x(int Wo) { return (Wo + STEM_TW - 1) / STEM_TW; }

}  // namespace zly
