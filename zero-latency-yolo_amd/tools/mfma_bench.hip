// mfma_bench.hip -- what would fp8 buy on the FLOP-dense layers?  (BASELINE configs[4]: "fp8 weights ... CDNA4 fp8 MFMA".)  Stand-alone
// microbenchmark, never part of libzly.so.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 zero-latency-yolo_amd/tools/mfma_bench.hip -o zero-latency-yolo_amd/_build/mfma_bench && ./zero-latency-yolo_amd/_build/mfma_bench
//
// Two questions, each answered for bf16 (v_mfma_f32_16x16x32_bf16), non-scaled fp8 (v_mfma_f32_16x16x32_fp8_fp8) and block-scaled fp8
// (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands, unit scales):
//   1. ISSUE: operands in registers, 9 x 3 independent accumulators per wave (the register blocking of conv3x3_ps_kernel), one and two
//      waves per SIMD, every CU busy: the matrix pipe's rate per dtype.
//   2. LDS-FED: the inner loop of conv3x3_ps_kernel for the Detect P3 stem (64 -> 144, NCT = 9 weight tiles and NPT = 3 pixel tiles per
//      k-step, fragments read from LDS with ds_read_b128, one barrier per k-step, two workgroups per CU): the same MACs with bf16
//      fragments (12 x 1 KiB reads per 27 MFMAs of K = 32) and with fp8 fragments (K = 128 per MFMA: 12 x 2 KiB reads per 27 MFMAs that do
//      4 x the MACs, i.e. half the LDS bytes per MAC).  No global memory traffic: this isolates pipe + LDS.
// Output: TFLOP/s per variant (2 * MACs), and the ratio to bf16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

enum { BF16 = 0, FP8 = 1, FP8S = 2, BF16K16 = 3 };          // BF16K16: v_mfma_f32_16x16x16_bf16 (the 4-bf16-per-lane form the 16-channel layers use)

template <int KIND> __device__ __forceinline__ f32x4 mma(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, f32x4 c)
{
    if constexpr (KIND == BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), c, 0, 0, 0);
    else if constexpr (KIND == BF16K16) {
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        const s16x4 sa = {(short)a0[0], (short)(a0[0] >> 16), (short)a0[1], (short)(a0[1] >> 16)}, sb = {(short)b0[0], (short)(b0[0] >> 16), (short)b0[1], (short)(b0[1] >> 16)};
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, c, 0, 0, 0);
    }
    else if constexpr (KIND == FP8) {
        // K = 32 fp8 values per MFMA: 8 bytes per lane per operand
        const long la = (long)a0[0] | ((long)a0[1] << 32), lb = (long)b0[0] | ((long)b0[1] << 32);
        return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c, 0, 0, 0);
    } else {
        i32x8 a, b;
        a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3]; a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
        b[0] = b0[0]; b[1] = b0[1]; b[2] = b0[2]; b[3] = b0[3]; b[4] = b1[0]; b[5] = b1[1]; b[6] = b1[2]; b[7] = b1[3];
        return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /* A: fp8 e4m3 */, 0 /* B: fp8 e4m3 */, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
}
template <int KIND> constexpr int kdepth() { return KIND == FP8S ? 128 : 32; }

// 1. issue rate: 27 accumulators, operands in registers
template <int KIND>
__global__ __launch_bounds__(256) void issue_kernel(float* out, int iters)
{
    u32x4 a0, a1, b0, b1;
    for (int j = 0; j < 4; ++j) { a0[j] = 0x38383838u + threadIdx.x; a1[j] = 0x3c3c3c3cu; b0[j] = 0x38403840u + j; b1[j] = 0x30303030u; }
    f32x4 acc[27];
    for (int i = 0; i < 27; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 27; ++i) acc[i] = mma<KIND>(a0, a1, b0, b1, acc[i]);
    }
    float s = 0.f;
    for (int i = 0; i < 27; ++i) s += acc[i][0] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

// 2. the ps-kernel inner loop: per k-step 9 A fragments (ring slot) + 3 B fragments (patch) from LDS, 27 MFMAs, one barrier
template <int KIND>
__global__ __launch_bounds__(256, 2) void ldsfed_kernel(float* out, int ksteps, int tiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int FB = KIND == FP8S ? 32 : (KIND == FP8 ? 8 : 16);       // fragment bytes per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int u = threadIdx.x; u < 72 * 1024 / 16; u += 256) reinterpret_cast<u32x4*>(smem)[u] = u32x4{0x38383838u + (unsigned)u, 0x3c3c3c3cu, 0x38403840u, 0x30303030u};
    __syncthreads();
    f32x4 acc[27];
    float s = 0.f;
    for (int tl = 0; tl < tiles; ++tl) {
        for (int i = 0; i < 27; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int k = 0; k < ksteps; ++k) {
            __syncthreads();
            const unsigned char* slot = smem + 36 * 1024 + (k & 3) * (9 * 64 * FB) + lane * FB;
            const unsigned char* px = smem + ((wave * 3 * 16 + (lane & 15)) * 160 + (lane >> 4) * FB + (k % 9) * 160) % (34 * 1024);
            u32x4 wa[9], wb[9], xa[3], xb[3];
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                if constexpr (KIND == FP8) { wa[c] = u32x4{reinterpret_cast<const unsigned*>(slot + c * 64 * FB)[0], reinterpret_cast<const unsigned*>(slot + c * 64 * FB)[1], 0u, 0u}; wb[c] = wa[c]; }
                else { wa[c] = *reinterpret_cast<const u32x4*>(slot + c * 64 * FB); wb[c] = KIND == FP8S ? *reinterpret_cast<const u32x4*>(slot + c * 64 * FB + 16) : wa[c]; }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if constexpr (KIND == FP8) { xa[i] = u32x4{reinterpret_cast<const unsigned*>(px + i * 16 * 160)[0], reinterpret_cast<const unsigned*>(px + i * 16 * 160)[1], 0u, 0u}; xb[i] = xa[i]; }
                else { xa[i] = *reinterpret_cast<const u32x4*>(px + i * 16 * 160); xb[i] = KIND == FP8S ? *reinterpret_cast<const u32x4*>(px + i * 16 * 160 + 16) : xa[i]; }
            }
#pragma unroll
            for (int c = 0; c < 9; ++c)
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[c * 3 + i] = mma<KIND>(wa[c], wb[c], xa[i], xb[i], acc[c * 3 + i]);
        }
        for (int i = 0; i < 27; ++i) s += acc[i][0];
    }
    if (s == 123.456f) out[0] = s;
}

template <typename F> static float time_ms(F&& launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();                       // warm-up (clock ramp)
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main()
{
    float* d; hipMalloc((void**)&d, 64);
    const char* names[3] = {"bf16  16x16x32           ", "fp8   16x16x32 (unscaled)", "fp8   16x16x128 f8f6f4   "};
    printf("1. issue rate, operands in registers, 27 accumulators per wave, 256 CUs\n");
    double base[2] = {0, 0};
    for (int wps = 1; wps <= 2; ++wps) {
        const int iters = 4000, grid = 256 * wps;
        float ms[3];
        ms[0] = time_ms([&] { hipLaunchKernelGGL(issue_kernel<BF16>, dim3(grid), dim3(256), 0, 0, d, iters); });
        ms[1] = time_ms([&] { hipLaunchKernelGGL(issue_kernel<FP8>, dim3(grid), dim3(256), 0, 0, d, iters); });
        ms[2] = time_ms([&] { hipLaunchKernelGGL(issue_kernel<FP8S>, dim3(grid), dim3(256), 0, 0, d, iters); });
        {
            const float m16 = time_ms([&] { hipLaunchKernelGGL(issue_kernel<BF16K16>, dim3(grid), dim3(256), 0, 0, d, iters); });
            printf("   %d wave(s)/SIMD  bf16  16x16x16            %8.1f TFLOP/s  (%.2f x the time of a 16x16x32 per instruction)\n", wps, 2.0 * grid * 4 * (double)iters * 27 * 16 * 16 * 16 / (m16 * 1e-3) / 1e12, m16 / ms[0]);
        }
        for (int k = 0; k < 3; ++k) {
            const double flops = 2.0 * grid * 4 * (double)iters * 27 * 16 * 16 * (k == 2 ? 128 : 32);
            const double tf = flops / (ms[k] * 1e-3) / 1e12;
            if (k == 0) base[wps - 1] = tf;
            printf("   %d wave(s)/SIMD  %s %8.1f TFLOP/s  (%.2f x bf16)\n", wps, names[k], tf, tf / base[wps - 1]);
        }
    }
    printf("2. conv3x3_ps_kernel inner loop (P3 stem: 9 weight + 3 pixel fragments from LDS per k-step, 27 MFMAs, one barrier), 2 workgroups per CU\n");
    hipFuncSetAttribute((const void*)ldsfed_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipFuncSetAttribute((const void*)ldsfed_kernel<FP8>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipFuncSetAttribute((const void*)ldsfed_kernel<FP8S>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    {
        // same MACs in all three: 18 k-steps of 32 channels per tile for the K = 32 forms; the K = 128 form needs 18 / 4 -> 5 (rounded up) k-steps
        const int tiles = 40, grid = 512;
        float ms[3];
        ms[0] = time_ms([&] { hipLaunchKernelGGL(ldsfed_kernel<BF16>, dim3(grid), dim3(256), 72 * 1024, 0, d, 18, tiles); });
        ms[1] = time_ms([&] { hipLaunchKernelGGL(ldsfed_kernel<FP8>, dim3(grid), dim3(256), 72 * 1024, 0, d, 18, tiles); });
        ms[2] = time_ms([&] { hipLaunchKernelGGL(ldsfed_kernel<FP8S>, dim3(grid), dim3(256), 72 * 1024, 0, d, 5, tiles); });
        const double macs = (double)grid * tiles * 4 * 18 * 27 * 16 * 16 * 32;      // algorithmic MACs of the bf16 loop (the f8f6f4 loop does 5/4.5 of them)
        for (int k = 0; k < 3; ++k) {
            const double tf = 2.0 * macs / (ms[k] * 1e-3) / 1e12;
            printf("   %s %8.3f ms  %8.1f TFLOP/s algorithmic  (%.2f x bf16)\n", names[k], ms[k], tf, ms[0] / ms[k]);
        }
    }
    hipFree(d);
    return 0;
}
