// grid_barrier_bench.hip -- what does a software grid barrier cost on MI355X?  (Feasibility of running the ~30 small-map layers of the batch-1
// latency path as ONE persistent kernel: a dependent launch costs ~4.3 us there)  Stand-alone, never part of libzly.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 zero-latency-yolo_amd/tools/grid_barrier_bench.hip -o zero-latency-yolo_amd/_build/grid_barrier_bench
// Every wait is BOUNDED (max_spin polls, then an abort flag that releases everybody): a barrier that cannot complete ends the kernel instead
// of hanging the GPU.  Variants: all workgroups (8 XCDs: release/acquire fences at agent scope), and only the workgroups of ONE XCD
// (blockIdx % 8 == 0 by the round-robin dispatch order, checked against the XCC_ID hardware register) with the same fences.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* ctr, unsigned* abort_flag, unsigned* xcc_seen, int nbar, int max_spin, int one_xcd, int participants, float* sink)
{
    if (threadIdx.x == 0) xcc_seen[blockIdx.x] = xcc_id();
    if (one_xcd && (blockIdx.x & 7) != 0) return;
    float acc = (float)threadIdx.x;
    for (int i = 0; i < nbar; ++i) {
        acc = acc * 1.0001f + 1.0f;                       // a token amount of work
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();                              // release: this workgroup's stores are visible device-wide
            atomicAdd(ctr, 1u);
            const unsigned target = (unsigned)(i + 1) * (unsigned)participants;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins >= max_spin || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(abort_flag, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __threadfence();                              // acquire
        }
        __syncthreads();
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (acc == 12345.678f) *sink = acc;
}

int main()
{
    unsigned *ctr, *flag, *xcc; float* sink;
    hipMalloc((void**)&ctr, 4); hipMalloc((void**)&flag, 4); hipMalloc((void**)&xcc, 4096 * 4); hipMalloc((void**)&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nbar = 200;
    for (int one_xcd = 0; one_xcd <= 1; ++one_xcd)
        for (int grid : {32, 64, 128, 256, 512}) {
            if (one_xcd && grid < 64) continue;
            const int participants = one_xcd ? grid / 8 : grid;
            float best = 1e9f; unsigned aborted = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(ctr, 0, 4); hipMemset(flag, 0, 4);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(256), 0, 0, ctr, flag, xcc, nbar, 200000, one_xcd, participants, sink);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
                hipMemcpy(&aborted, flag, 4, hipMemcpyDeviceToHost);
            }
            std::vector<unsigned> h(grid); hipMemcpy(h.data(), xcc, grid * 4, hipMemcpyDeviceToHost);
            int same = 0, part = 0; for (int b = 0; b < grid; ++b) if (!one_xcd || (b & 7) == 0) { ++part; same += h[b] == h[0]; }
            printf("%s grid %3d (%3d participating workgroups, %3d of them on XCC %u)%s: %.2f us per barrier (%d barriers, kernel %.1f us)\n",
                   one_xcd ? "one XCD " : "all XCDs", grid, part, same, h[0], aborted ? "  ABORTED" : "", (best * 1e3 - 6.0) / nbar, nbar, best * 1e3);
        }
    return 0;
}
