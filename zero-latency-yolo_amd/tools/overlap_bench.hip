// overlap_bench.hip -- what does a chain of dependent launches cost when OTHER chains run beside it?  Stand-alone, never part of libzly.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 zero-latency-yolo_amd/tools/overlap_bench.hip -o zero-latency-yolo_amd/_build/overlap_bench
// Background (profiles/r04_launch_ablation_3engines.txt): with three engines' chains in flight, leaving ONE ~10 us launch out of every
// chain shortens the 64-frame step by ~20 us -- twice the launch's isolated duration -- while a 45 us launch saves ~45.  This bench rebuilds that
// situation from synthetic kernels to see which resource the short launches are queueing for:
//   chain  = 40 dependent launches of `work_kernel` (grid G x 256 threads, LDS bytes L, ~T us of dependent FMAs, reads its predecessor's
//            output, writes its own), replayed as ONE hipGraph per stream, R replays back to back;
//   beside = nothing | the same chain on 1 / 2 more streams | a chain of chip-filling ALU kernels | a chain of streaming (memory-bound) kernels.
// Reported: us per launch of the measured chain (its own events), alone and beside the others.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(256) void work_kernel(const float4* __restrict__ in, float4* __restrict__ out, int iters, int lds_touch)
{
    extern __shared__ float sm[];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 v = in[i];
    if (lds_touch) { sm[threadIdx.x] = v.x; __syncthreads(); v.y += sm[255 - threadIdx.x]; }
    float a = v.x, b = v.y;
    for (int k = 0; k < iters; ++k) { a = a * 1.0001f + b; b = b * 0.9999f + a; }      // dependent chain: latency, not throughput
    v.x = a; v.y = b;
    out[i] = v;
}

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = in[i]; v.x += 1.0f; out[i] = v; }
}

struct Chain {
    hipStream_t st; hipGraphExec_t ge; hipEvent_t e0, e1;
    float4 *a, *b;
};

// kind 0: work_kernel chain (grid, lds, iters); kind 1: streaming chain (bytes per launch)
static Chain make_chain(int kind, int nlaunch, int grid, int lds, int iters, size_t bytes)
{
    Chain c{};
    (void)hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking);
    (void)hipEventCreate(&c.e0); (void)hipEventCreate(&c.e1);
    const size_t n4 = kind == 0 ? (size_t)grid * 256 : bytes / 16;
    (void)hipMalloc((void**)&c.a, n4 * 16); (void)hipMalloc((void**)&c.b, n4 * 16);
    (void)hipMemset(c.a, 0, n4 * 16); (void)hipMemset(c.b, 0, n4 * 16);
    if (kind == 0 && lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)work_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipGraph_t gr;
    (void)hipStreamBeginCapture(c.st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nlaunch; ++i) {
        const float4* in = (i & 1) ? c.b : c.a; float4* out = (i & 1) ? c.a : c.b;
        if (kind == 0) hipLaunchKernelGGL(work_kernel, dim3(grid), dim3(256), (size_t)(lds < 1024 ? 1024 : lds), c.st, in, out, iters, lds > 0 ? 1 : 0);
        else hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, c.st, in, out, n4);
    }
    (void)hipStreamEndCapture(c.st, &gr);
    (void)hipGraphInstantiate(&c.ge, gr, nullptr, nullptr, 0);
    (void)hipGraphDestroy(gr);
    return c;
}
static void free_chain(Chain& c) { (void)hipGraphExecDestroy(c.ge); (void)hipStreamDestroy(c.st); (void)hipEventDestroy(c.e0); (void)hipEventDestroy(c.e1); (void)hipFree(c.a); (void)hipFree(c.b); }

// R replays of every chain, all streams started together; returns the measured chain's (index 0) us per launch, best of `reps`
static float run_together(std::vector<Chain>& cs, int nlaunch, int R, int reps, float* others_us = nullptr)
{
    float best = 1e9f, obest = 1e9f;
    for (int rep = 0; rep < reps; ++rep) {
        for (Chain& c : cs) (void)hipEventRecord(c.e0, c.st);
        for (int r = 0; r < R; ++r) for (Chain& c : cs) (void)hipGraphLaunch(c.ge, c.st);
        for (Chain& c : cs) (void)hipEventRecord(c.e1, c.st);
        for (Chain& c : cs) (void)hipEventSynchronize(c.e1);
        float ms; (void)hipEventElapsedTime(&ms, cs[0].e0, cs[0].e1);
        if (ms < best) best = ms;
        if (cs.size() > 1) { (void)hipEventElapsedTime(&ms, cs[1].e0, cs[1].e1); if (ms < obest) obest = ms; }
    }
    if (others_us) *others_us = obest * 1e3f / (float)(nlaunch * R);
    return best * 1e3f / (float)(nlaunch * R);
}

int main()
{
    const int N = 40, R = 8, reps = 6;
    printf("# chain = %d dependent launches per graph, %d replays back to back; us per launch of the measured chain (best of %d)\n", N, R, reps);
    struct Shape { const char* name; int grid, lds, iters; };
    const Shape shapes[] = {
        {"empty-ish: 256 WGs, no LDS, 0 FMAs", 256, 0, 0},
        {"512 WGs, no LDS, ~4 us of FMAs", 512, 0, 600},
        {"512 WGs, 60 KB LDS (2 per CU), ~4 us of FMAs", 512, 60 * 1024, 600},
        {"512 WGs, 100 KB LDS (1 per CU), ~4 us of FMAs", 512, 100 * 1024, 600},
        {"2048 WGs, 60 KB LDS, ~4 us of FMAs (4 rounds of workgroups)", 2048, 60 * 1024, 600},
    };
    for (const Shape& s : shapes) {
        std::vector<Chain> one{make_chain(0, N, s.grid, s.lds, s.iters, 0)};
        const float alone = run_together(one, N, R, reps);
        std::vector<Chain> two{one[0], make_chain(0, N, s.grid, s.lds, s.iters, 0)};
        const float with1 = run_together(two, N, R, reps);
        std::vector<Chain> three{two[0], two[1], make_chain(0, N, s.grid, s.lds, s.iters, 0)};
        const float with2 = run_together(three, N, R, reps);
        printf("%-62s alone %6.2f | beside 1 twin %6.2f | beside 2 twins %6.2f   (per launch of ONE chain; perfect overlap = alone, none = x2 / x3)\n", s.name, alone, with1, with2);
        // beside a streaming chain (64 MB read + 64 MB written per launch: ~25 us each)
        std::vector<Chain> mix{three[0], make_chain(1, 10, 0, 0, 0, (size_t)64 << 20)};
        float o = 0;
        const float withs = run_together(mix, N, R, reps, &o);
        printf("%-62s beside a streaming chain (64 MB in + out per launch): %6.2f  (the streaming chain alone-equivalent per launch: %.1f us x %d launches)\n", "", withs, o * N * R / (10.0f * R), 10);
        free_chain(mix[1]); free_chain(three[2]); free_chain(two[1]); free_chain(one[0]);
    }
    return 0;
}
