// valu_bench.hip -- issue rates of the VALU instructions the conv epilogues are made of (bias + SiLU + convert), MI355X.  Stand-alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 zero-latency-yolo_amd/tools/valu_bench.hip -o zero-latency-yolo_amd/_build/valu_bench
// Every kernel issues the same number of INDEPENDENT instructions of one kind per wave (8 register chains, round robin); reported: cycles per
// wave-instruction per SIMD at 1 / 2 / 4 waves per SIMD (s_memtime around the loop, mean over the waves of the launch) -- 4 cycles = full rate
// for a 64-lane wave on a 16-lane SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void valu_kernel(float* out, unsigned long long* cyc, int iters)
{
    float r[8]; f32x2 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { r[i] = 1.0f + 0.001f * (float)(threadIdx.x + i); q[i] = f32x2{r[i], r[i] * 0.5f}; }
    const float k = 0.999f; const f32x2 k2 = {0.999f, 1.001f};
    int ri[8]; const int ki = 3 + (int)(threadIdx.x & 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) ri[i] = (int)threadIdx.x + i;
    const unsigned long long msk = 0x5555555555555555ull ^ (unsigned long long)blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(k));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[i]) : "v"(k2));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[i]) : "v"(k2));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q[i]) : "v"(k2));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(k));
#define CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "s"(msk));
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ri[i]) : "v"(ki));
#define MUL24(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(ri[i]) : "v"(ki));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ri[i]) : "v"(ki));
#define CVTUB(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r[i]));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
#define PKMAXI(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(ri[i]) : "v"(ki));
#define SILU(i) { float t_; asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %1\n\tv_exp_f32 %0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_rcp_f32 %0, %0\n\tv_mul_f32 %1, %1, %0" : "=&v"(t_), "+v"(r[i])); }
        if (KIND == 0) { REP8(MUL) REP8(MUL) }
        else if (KIND == 1) { REP8(FMA) REP8(FMA) }
        else if (KIND == 2) { REP8(EXP) REP8(EXP) }
        else if (KIND == 3) { REP8(RCP) REP8(RCP) }
        else if (KIND == 4) { REP8(PKMUL) REP8(PKMUL) }
        else if (KIND == 5) { REP8(PKADD) REP8(PKADD) }
        else if (KIND == 6) { REP8(PKFMA) REP8(PKFMA) }
        else if (KIND == 7) { REP8(CVT) REP8(CVT) }
        else if (KIND == 8) { REP8(CND) REP8(CND) }
        else if (KIND == 9) { REP8(SILU) REP8(SILU) }
        else if (KIND == 10) { REP8(CND64) REP8(CND64) }
        else if (KIND == 11) { REP8(ADDU) REP8(ADDU) }
        else if (KIND == 12) { REP8(MUL24) REP8(MUL24) }
        else if (KIND == 13) { REP8(MULLO) REP8(MULLO) }
        else if (KIND == 14) { REP8(CVTUB) REP8(CVTUB) }
        else if (KIND == 15) { REP8(MAXF) REP8(MAXF) }
        else if (KIND == 16) { REP8(PKMAXI) REP8(PKMAXI) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i] + q[i][0] + q[i][1] + (float)ri[i];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_iter_insts)
{
    float* out; unsigned long long* cyc;
    (void)hipMalloc((void**)&out, 4); (void)hipMalloc((void**)&cyc, 8 * 4096 * 4);
    const int iters = 2000;
    printf("%-44s", name);
    for (int wps : {1, 2, 4}) {
        // 256 CUs x 4 SIMDs: wps waves per SIMD = wps workgroups of 4 waves per CU
        const int grid = 256 * wps;
        hipLaunchKernelGGL(valu_kernel<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(valu_kernel<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)grid * 4);
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        const double per_wave_inst = s / h.size() / ((double)iters * per_iter_insts);      // cycles of wave lifetime per instruction
        printf("  %d wave(s)/SIMD: %6.2f cyc/inst/wave = %5.2f cyc/inst/SIMD", wps, per_wave_inst, per_wave_inst / wps);
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    printf("# unit = s_memtime ticks (not necessarily core cycles): compare rows with each other\n");
    run<0>("v_mul_f32", 16);
    run<1>("v_fma_f32", 16);
    run<2>("v_exp_f32", 16);
    run<3>("v_rcp_f32", 16);
    run<4>("v_pk_mul_f32 (2 results per lane)", 16);
    run<5>("v_pk_add_f32 (2 results per lane)", 16);
    run<6>("v_pk_fma_f32 (2 results per lane)", 16);
    run<7>("v_cvt_pk_bf16_f32", 16);
    run<8>("v_cndmask_b32", 16);
    run<9>("SiLU chain: mul, exp, add, rcp, mul (5 inst)", 16 * 5);
    run<10>("v_cndmask_b32_e64 (SGPR-pair mask)", 16);
    run<11>("v_add_u32", 16);
    run<12>("v_mul_i32_i24", 16);
    run<13>("v_mul_lo_u32", 16);
    run<14>("v_cvt_f32_ubyte1", 16);
    run<15>("v_max_f32", 16);
    run<16>("v_pk_max_i16", 16);
    return 0;
}
