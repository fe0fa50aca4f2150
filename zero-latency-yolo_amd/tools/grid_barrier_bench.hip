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


// XCD-hierarchical form (MI355X_MICROARCH.md price list, row "barrier-xcd"): workgroups are grouped by blockIdx % 8 (the round-robin dispatch puts such
// a group on one XCD -- for speed only, nothing depends on it); per group a counter on a line of its own; the group's LAST arriver is its leader: release
// fence (writes the XCD's L2 back), add to the top counter, poll it (relaxed sc1 loads + s_sleep) until all groups are in, acquire fence, publish the
// group's generation word; every other workgroup has drained its stores into the shared L2 (s_waitcnt vmcnt(0)) before its add, polls the generation word
// and ends with an acquire fence (L1 invalidate).  Every spin is bounded; an abort flag releases everybody.
struct XcdBar { unsigned cnt[8][32]; unsigned gen[8][32]; unsigned top[32]; };      // each word on a 128-byte line of its own
__global__ __launch_bounds__(256) void barrier_xcd_kernel(XcdBar* bar, unsigned* abort_flag, int nbar, int max_spin, int grid, float* sink)
{
    const int g = blockIdx.x & 7;
    const int ngroups = grid < 8 ? grid : 8;
    const int members = (grid - g + 7) >> 3;
    float acc = (float)threadIdx.x;
    for (int i = 0; i < nbar; ++i) {
        acc = acc * 1.0001f + 1.0f;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned epoch = (unsigned)(i + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned old = __hip_atomic_fetch_add(&bar->cnt[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            bool aborted = false;
            if (old == epoch * (unsigned)members - 1u) {                // last arriver of this group: the leader
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&bar->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&bar->top[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * (unsigned)ngroups) {
                    if (++spins >= max_spin || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(abort_flag, 1u); aborted = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(&bar->gen[g][0], aborted ? 0xffffffffu : epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(&bar->gen[g][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                    if (++spins >= max_spin || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(abort_flag, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (acc == 12345.678f) *sink = acc;
}

// an empty kernel (and one that touches its arguments) as a chain of dependent launches, eager and as a hipGraph: the per-launch floor of the batch-1 path
struct FatArgs { const void* p[16]; int v[16]; };
__global__ __launch_bounds__(256) void empty_kernel() {}
__global__ __launch_bounds__(256) void touch_kernel(const FatArgs a, float* out) { if (threadIdx.x == 0 && a.v[3] == 12345) out[blockIdx.x] = (float)a.v[0]; }
__global__ __launch_bounds__(256) void rw_kernel(const float* in, float* out, int n)     // reads what its predecessor wrote: a real dependency through memory
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0f;
}
static void chain_bench(const char* name, int grid, int kind, float* bufa, float* bufb)
{
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int N = 40;
    FatArgs fa{}; float* out = bufa;
    auto launch_all = [&]() {
        for (int i = 0; i < N; ++i) {
            if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st);
            else if (kind == 1) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, st, fa, out);
            else hipLaunchKernelGGL(rw_kernel, dim3(grid), dim3(256), 0, st, (i & 1) ? bufb : bufa, (i & 1) ? bufa : bufb, grid * 256);
        }
    };
    hipGraph_t gr; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    launch_all();
    (void)hipStreamEndCapture(st, &gr);
    (void)hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best_g = 1e9f, best_e = 1e9f, ms;
    for (int rep = 0; rep < 20; ++rep) {
        (void)hipEventRecord(e0, st); (void)hipGraphLaunch(ge, st); (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best_g) best_g = ms;
    }
    for (int rep = 0; rep < 20; ++rep) {
        (void)hipEventRecord(e0, st); launch_all(); (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best_e) best_e = ms;
    }
    printf("chain of %d dependent launches, %-34s grid %4d: %.2f us per launch as a graph (%.1f us per replay), %.2f us eager\n", N, name, grid, best_g * 1e3 / N, best_g * 1e3, best_e * 1e3 / N);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(gr); (void)hipStreamDestroy(st);
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
    XcdBar* xb; hipMalloc((void**)&xb, sizeof(XcdBar));
    for (int grid : {8, 16, 32, 64, 128, 256, 512}) {
        float best = 1e9f; unsigned aborted = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(xb, 0, sizeof(XcdBar)); hipMemset(flag, 0, 4);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(barrier_xcd_kernel, dim3(grid), dim3(256), 0, 0, xb, flag, nbar, 200000, grid, sink);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            hipMemcpy(&aborted, flag, 4, hipMemcpyDeviceToHost);
        }
        printf("XCD-hierarchical grid %3d%s: %.2f us per barrier (%d barriers, kernel %.1f us)\n", grid, aborted ? "  ABORTED" : "", (best * 1e3 - 6.0) / nbar, nbar, best * 1e3);
    }
    float *ba, *bb; hipMalloc((void**)&ba, 1024 * 256 * 4); hipMalloc((void**)&bb, 1024 * 256 * 4); hipMemset(ba, 0, 1024 * 256 * 4); hipMemset(bb, 0, 1024 * 256 * 4);
    for (int grid : {1, 64, 256, 1024}) {
        chain_bench("empty kernel", grid, 0, ba, bb);
        chain_bench("kernel reading a 192-byte kernarg", grid, 1, ba, bb);
        chain_bench("kernel reading its predecessor's output", grid, 2, ba, bb);
    }
    return 0;
}
