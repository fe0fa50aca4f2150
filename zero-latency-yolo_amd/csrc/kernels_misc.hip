// kernels_misc.hip -- the non-GEMM stages of the detect path on gfx950: stand-alone preprocess (parity entry
// point / fp32 engine) and the SPPF max-pools.  (Upsample+Concat is fused into the consumer conv, the Detect
// tail lives in kernels_head.hip, preprocess+stem in kernels_stem.hip.)  HBM/LDS-bound element work:
// 16-byte accesses, no MFMA.
#include "zly_internal.h"

namespace zly {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ------------------------------------------------------------------------------------------------
// preprocess: stretch nearest-neighbour resize + BGR->RGB + /255
// reference OnnxInferenceEngine::preProcess, src/inference/onnx_engine.cpp:649-700:
//   scale_w = float(w)/tw, scale_h = float(h)/th                       (:673-674)
//   src_y = min(int(y*scale_h), h-1); src_x = min(int(x*scale_w), w-1)  (:681-682)
//   out[c][y][x] = src[(src_y*w + src_x)*3 + (2-c)] / 255.0f            (:685,:693)
// Every operation is a single IEEE fp32 op, so the result is bit-identical to the CPU oracle.
// Two outputs: the engine's NHWC tensor padded to 8 channels (one 16-byte store per pixel in bf16),
// and optionally the reference's planar fp32 [3][th][tw] layout for the parity entry point.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ src, const FrameDesc* __restrict__ desc,
                                                         T* __restrict__ out8, float* __restrict__ out_nchw, int tw, int th)
{
    const int f = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= tw * th) return;
    const int y = idx / tw, x = idx - y * tw;
    const FrameDesc d = desc[f];
    const float scale_w = (float)d.w / (float)tw;
    const float scale_h = (float)d.h / (float)th;
    int sy = (int)((float)y * scale_h); if (sy > d.h - 1) sy = d.h - 1;
    int sx = (int)((float)x * scale_w); if (sx > d.w - 1) sx = d.w - 1;
    const uint8_t* px = src + d.src_off + ((size_t)sy * d.w + sx) * 3;
    const float b = (float)px[0] / 255.0f, g = (float)px[1] / 255.0f, r = (float)px[2] / 255.0f;
    if (out8) {
        T* o = out8 + ((size_t)f * th * tw + idx) * 8;
        o[0] = (T)r; o[1] = (T)g; o[2] = (T)b; o[3] = (T)0.f;
        o[4] = (T)0.f; o[5] = (T)0.f; o[6] = (T)0.f; o[7] = (T)0.f;
    }
    if (out_nchw) {
        float* o = out_nchw + (size_t)f * 3 * th * tw;
        o[idx] = r; o[(size_t)th * tw + idx] = g; o[(size_t)2 * th * tw + idx] = b;
    }
}

hipError_t launch_preprocess(int dtype, const uint8_t* src, const FrameDesc* desc, int n,
                             void* out_nhwc8, float* out_nchw_f32, int tw, int th, hipStream_t s)
{
    dim3 grid((tw * th + 255) / 256, n);
    if (dtype == ZLY_DTYPE_BF16)
        hipLaunchKernelGGL(preprocess_kernel<bf16_t>, grid, dim3(256), 0, s, src, desc, (bf16_t*)out_nhwc8, out_nchw_f32, tw, th);
    else
        hipLaunchKernelGGL(preprocess_kernel<float>, grid, dim3(256), 0, s, src, desc, (float*)out_nhwc8, out_nchw_f32, tw, th);
    return hipGetLastError();
}

// fp32 planar [n][3][th][tw] (the "images" tensor of onnx_engine.cpp:560-569) -> engine NHWC8
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc8_kernel(const float* __restrict__ in, T* __restrict__ out8, int hw)
{
    const int f = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= hw) return;
    const float* i = in + (size_t)f * 3 * hw;
    T* o = out8 + ((size_t)f * hw + idx) * 8;
    o[0] = (T)i[idx]; o[1] = (T)i[(size_t)hw + idx]; o[2] = (T)i[(size_t)2 * hw + idx]; o[3] = (T)0.f;
    o[4] = (T)0.f; o[5] = (T)0.f; o[6] = (T)0.f; o[7] = (T)0.f;
}

hipError_t launch_nchw_to_nhwc8(int dtype, const float* in_nchw, void* out_nhwc8, int n, int tw, int th, hipStream_t s)
{
    dim3 grid((tw * th + 255) / 256, n);
    if (dtype == ZLY_DTYPE_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc8_kernel<bf16_t>, grid, dim3(256), 0, s, in_nchw, (bf16_t*)out_nhwc8, tw * th);
    else
        hipLaunchKernelGGL(nchw_to_nhwc8_kernel<float>, grid, dim3(256), 0, s, in_nchw, (float*)out_nhwc8, tw * th);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// SPPF: three chained 5x5 stride-1 max-pools (pad 2, -inf outside).  k chained 5x5 pools are one max over
// a (4k+1)^2 window clipped to the map, so pool1/2/3 = windows of radius 2/4/6, and the max is separable.
// The source is channel block 0 ([0,c)) of the SPPF concat buffer; the three pools are written to channel
// blocks 1..3 of the same buffer, so the following 1x1 conv reads the 4c-channel concat without a Concat op.
// One workgroup = one frame x 8 channels, tile in LDS as fp32 (max is exact on bf16 values):
//   row pass   : every element gets its running row maxima of radius 2, 4 and 6 (13 LDS reads, cumulative)
//   column pass: radius-r column max over the radius-r row maxima (5 + 9 + 13 reads), stored straight out.
// Two barriers in total (the chained formulation needed seven).
// ------------------------------------------------------------------------------------------------
// 8 channels of one pixel as two float4 (one thread = one pixel: 16-byte global accesses, b128 LDS accesses)
struct Px8 { f32x4 lo, hi; };
__device__ __forceinline__ Px8 px_max(Px8 a, const Px8& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { a.lo[j] = fmaxf(a.lo[j], b.lo[j]); a.hi[j] = fmaxf(a.hi[j], b.hi[j]); }
    return a;
}
__device__ __forceinline__ Px8 load_px8(const bf16_t* p) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
    Px8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o.lo[j] = (float)v[j]; o.hi[j] = (float)v[4 + j]; }
    return o;
}
__device__ __forceinline__ Px8 load_px8(const float* p) {
    Px8 o; o.lo = *reinterpret_cast<const f32x4*>(p); o.hi = *reinterpret_cast<const f32x4*>(p + 4); return o;
}
__device__ __forceinline__ void store_px8(bf16_t* p, const Px8& v) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = (bf16_t)v.lo[j]; o[4 + j] = (bf16_t)v.hi[j]; }
    *reinterpret_cast<bf16x8*>(p) = o;
}
__device__ __forceinline__ void store_px8(float* p, const Px8& v) {
    *reinterpret_cast<f32x4*>(p) = v.lo; *reinterpret_cast<f32x4*>(p + 4) = v.hi;
}

// SPPF's three chained 5x5 max pools (stride 1, pad 2) of an 8-channel slice of one frame, in LDS.  pool(pool(x)) is the 9x9 pool and the third
// the 13x13 one, so each stage is a separable radius-2 pool of the previous stage's result: 4 + 4 neighbour reads per pixel and stage (round 3;
// the first version read radius-6 rows and columns of the input: 12 + 36).  max is exact in any order: same bits.
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_kernel(T* __restrict__ buf, int cs, int c, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int hw = H * W;
    Px8* A = reinterpret_cast<Px8*>(lds);       // current stage's input
    Px8* R = A + hw;                            // its row maxima
    Px8* B = R + hw;                            // its output = next stage's input
    const int f = blockIdx.y, c0 = blockIdx.x * 8;
    T* base = buf + (size_t)f * hw * cs + c0;
    for (int px = threadIdx.x; px < hw; px += 256) A[px] = load_px8(base + (size_t)px * cs);
    __syncthreads();
    for (int stage = 1; stage <= 3; ++stage) {
        for (int px = threadIdx.x; px < hw; px += 256) {
            const int y = px / W, x = px - y * W;
            const Px8* row = A + y * W;
            Px8 m = row[x];
#pragma unroll
            for (int d = 1; d <= 2; ++d) {
                if (x - d >= 0) m = px_max(m, row[x - d]);
                if (x + d < W) m = px_max(m, row[x + d]);
            }
            R[px] = m;
        }
        __syncthreads();
        for (int px = threadIdx.x; px < hw; px += 256) {
            const int y = px / W, x = px - y * W;
            Px8 m = R[px];
#pragma unroll
            for (int d = 1; d <= 2; ++d) {
                if (y - d >= 0) m = px_max(m, R[(y - d) * W + x]);
                if (y + d < H) m = px_max(m, R[(y + d) * W + x]);
            }
            B[px] = m;
            store_px8(base + (size_t)px * cs + stage * c, m);
        }
        __syncthreads();
        Px8* t = A; A = B; B = t;
    }
}

// ------------------------------------------------------------------------------------------------
// The same three pools for maps of at most 16 x 16 pixels in bf16 (13 x 13 at 416 x 416: the headline), round 4.  sppf_pool_kernel above runs six
// LDS passes separated by workgroup barriers on fp32 records (13.4 us per launch at batch 64 for 11 MB of traffic; an ablated step is ~20 us shorter).
// Here: pool_k = (2k+1) x (2k+1) window of y with k = 2, 4, 6 = column window of radius k over the ROW window of radius k, and a row window of radius
// k + 2 is the radius-2 row window of the radius-k one.  A workgroup is 16 rows x 16 lanes = one frame's map for 8 channels, a thread = one pixel
// (16 bytes = 4 dwords of bf16 pairs):
//   * values go to the sortable-int16 domain (T(x) = x ^ ((x >> 15) & 0x7fff) per half-word, an involution): v_pk_max_i16, two channels per instruction,
//     exact -- max is a selection, the same bits come out as from the fp32 compare;
//   * the three ROW windows are built in registers with DPP row shifts (a DPP row IS a map row; lanes shifted in from outside the row, and the lanes /
//     rows beyond the map, hold the padding value 0x8000 = below every T(x));
//   * r2, r4, r6 go to LDS once, ONE barrier, and every pixel reads its column windows directly: 5 + 9 + 13 rows with the row index clamped into the map
//     (max is idempotent: a clamped duplicate changes nothing, so there is no bounds test);
//   * three 16-byte stores per pixel into the concat buffer, as before.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int pu32x4;
typedef __attribute__((ext_vector_type(8))) short ps16x8;
__device__ __forceinline__ pu32x4 pool_sortable(pu32x4 v)
{
    pu32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const unsigned sg = v[i] & 0x80008000u; r[i] = v[i] ^ (sg - (sg >> 15)); }     // per half-word: negative -> flip the low 15 bits (no borrow between the halves)
    return r;
}
__device__ __forceinline__ pu32x4 pool_max(pu32x4 a, pu32x4 b)
{
    return __builtin_bit_cast(pu32x4, __builtin_elementwise_max(__builtin_bit_cast(ps16x8, a), __builtin_bit_cast(ps16x8, b)));
}
template <int CTRL> __device__ __forceinline__ pu32x4 pool_row_shift(pu32x4 v)
{
    pu32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (unsigned)__builtin_amdgcn_update_dpp((int)0x80008000u, (int)v[i], CTRL, 0xf, 0xf, false);    // lanes without a source keep the padding value
    return r;
}
__device__ __forceinline__ pu32x4 pool_row5(pu32x4 v)           // radius-2 window along the DPP row (= map row)
{
    pu32x4 m = pool_max(v, pool_row_shift<0x111>(v));          // row_shr:1 -- lane x <- lane x - 1
    m = pool_max(m, pool_row_shift<0x112>(v));                 // row_shr:2
    m = pool_max(m, pool_row_shift<0x101>(v));                 // row_shl:1 -- lane x <- lane x + 1
    return pool_max(m, pool_row_shift<0x102>(v));              // row_shl:2
}
__global__ __launch_bounds__(256) void sppf_pool16_kernel(unsigned short* __restrict__ buf, int cs, int c, int H, int W)
{
    __shared__ __attribute__((aligned(16))) pu32x4 rows[3][16][16];       // [radius 2 / 4 / 6][map row][map column]: 12 KB
    const int x = threadIdx.x & 15, y = threadIdx.x >> 4;
    const int f = blockIdx.y, c0 = blockIdx.x * 8;
    const bool in = y < H && x < W;
    unsigned short* px = buf + ((size_t)f * H * W + (size_t)(in ? y * W + x : 0)) * cs + c0;
    const pu32x4 pad = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
    pu32x4 v = pad;
    if (in) v = pool_sortable(*reinterpret_cast<const pu32x4*>(px));
    const pu32x4 r2 = pool_row5(v), r4 = pool_row5(r2), r6 = pool_row5(r4);
    rows[0][y][x] = r2; rows[1][y][x] = r4; rows[2][y][x] = r6;
    __syncthreads();
    if (!in) return;
    const int ymax = H - 1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int rad = 2 * k + 2;
        pu32x4 m = rows[k][y][x];
#pragma unroll
        for (int d = 1; d <= rad; ++d) {
            m = pool_max(m, rows[k][max(y - d, 0)][x]);
            m = pool_max(m, rows[k][min(y + d, ymax)][x]);
        }
        *reinterpret_cast<pu32x4*>(px + (size_t)(k + 1) * c) = pool_sortable(m);
    }
}

hipError_t launch_sppf_pool(int dtype, void* buf, int cs, int c, int n, int H, int W, hipStream_t s, int six_pass)
{
    constexpr int SPPF_POOL16_MAX_BATCH = 16;
    // small maps in bf16: the one-barrier kernel (six_pass: the engine's ZLY_SPPF_POOL_LDS=1 switch, tests / A-B)
    // Small batches only: at batch 1 the launch is 6.8 us against 8.8 (one barrier instead of six on the latency path); at batch 64 it is 13.7 us against 16.5 in
    // isolation -- both kernels move the map in 16-byte pieces 1 KB apart (an 8-channel slice of a 512-channel NHWC buffer), which is what bounds them
    // there -- and the three-engine step is 0.8 % SLOWER with it (A/B, 3 alternating rounds), so the big batches keep the six-pass kernel.
    if (dtype == ZLY_DTYPE_BF16 && H <= 16 && W <= 16 && (c % 8) == 0 && (cs % 8) == 0 && !six_pass && n <= SPPF_POOL16_MAX_BATCH) {
        hipLaunchKernelGGL(sppf_pool16_kernel, dim3(c / 8, n), dim3(256), 0, s, (unsigned short*)buf, cs, c, H, W);
        return hipGetLastError();
    }

    const size_t lds = (size_t)H * W * 8 * sizeof(float) * 3;
    if (lds > 160 * 1024 || (c % 8) != 0) return hipErrorInvalidValue;
    dim3 grid(c / 8, n);
    if (dtype == ZLY_DTYPE_BF16) {
        if (lds > 64 * 1024) hipFuncSetAttribute((const void*)sppf_pool_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(sppf_pool_kernel<bf16_t>, grid, dim3(256), lds, s, (bf16_t*)buf, cs, c, H, W);
    } else {
        if (lds > 64 * 1024) hipFuncSetAttribute((const void*)sppf_pool_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(sppf_pool_kernel<float>, grid, dim3(256), lds, s, (float*)buf, cs, c, H, W);
    }
    return hipGetLastError();
}

// debug/parity: channel slice of frame idx -> fp32 planar [C][H][W]
template <typename T>
__global__ __launch_bounds__(256) void tap_kernel(const T* __restrict__ in, int cs, int co, int C, int hw, float* __restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)C * hw) return;
    const int c = (int)(i / hw), px = (int)(i - (long)c * hw);
    out[i] = (float)in[(size_t)px * cs + co + c];
}

hipError_t launch_tap_to_nchw(int dtype, const void* in, int cs, int co, int C, int H, int W, int idx, float* out, hipStream_t s)
{
    const int hw = H * W;
    const long total = (long)C * hw;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == ZLY_DTYPE_BF16)
        hipLaunchKernelGGL(tap_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in + (size_t)idx * hw * cs, cs, co, C, hw, out);
    else if (dtype == ZLY_DTYPE_FP32)
        hipLaunchKernelGGL(tap_kernel<float>, grid, dim3(256), 0, s, (const float*)in + (size_t)idx * hw * cs, cs, co, C, hw, out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace zly
