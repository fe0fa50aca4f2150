// nhwc_bw.hip -- DIAGNOSTIC: how fast can the conv kernels' NHWC epilogue / fragment access pattern move bytes at all?
// Every conv kernel of the engine stores (and the direct kernels load) with the MFMA lane mapping: lane (p = lane & 15, kq = lane >> 4)
// touches 16 bytes at  pixel(p) * pitch + (tile pair) * 64 + kq * 16  -- per wave-instruction 16 pixels x 64 contiguous bytes, pixel
// pitch = channels * 2 bytes.  This measures that pattern against a plain coalesced copy, for stores and for loads, per channel count:
//   "mfma"   : the pattern above; the channel pairs of a pixel tile are written by `split` different waves (channel-split kernels) or by one
//   "linear" : the same bytes with consecutive lanes on consecutive 16-byte units (what a transposed epilogue through LDS would issue)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 zero-latency-yolo_amd/tools/nhwc_bw.hip -o zero-latency-yolo_amd/_build/nhwc_bw && ./zero-latency-yolo_amd/_build/nhwc_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// one wave = one 16-pixel tile x `pairs_per_wave` 64-byte channel pairs; waves of a workgroup take consecutive pixel tiles (split == 1) or the
// channel pairs of the SAME pixel tile (split == waves per pixel tile)
template <bool STORE>
__global__ __launch_bounds__(256) void mfma_pattern(u32x4* buf, int npx, int pitch16 /* pixel pitch in 16-byte units */, int pairs, int split, unsigned* sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, kq = lane >> 4;
    const int gw = blockIdx.x * 4 + wave;                       // global wave index
    const int tile = gw / split, part = gw - tile * split;      // pixel tile, channel part
    const int pairs_per_wave = pairs / split;
    const long px = (long)tile * 16 + p;
    if (px >= npx) return;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int c = 0; c < pairs_per_wave; ++c) {
        u32x4* q = buf + px * pitch16 + (part * pairs_per_wave + c) * 4 + kq;
        if (STORE) *q = u32x4{(unsigned)px, (unsigned)c, 3u, 4u};
        else { const u32x4 v = *q; acc[0] += v[0]; acc[1] ^= v[1]; acc[2] += v[2]; acc[3] ^= v[3]; }
    }
    if (!STORE && acc[0] + acc[1] + acc[2] + acc[3] == 0x12345u) *sink = 1;
}

template <bool STORE>
__global__ __launch_bounds__(256) void linear_pattern(u32x4* buf, long units, unsigned* sink)
{
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long)gridDim.x * 256) {
        if (STORE) buf[u] = u32x4{(unsigned)u, 2u, 3u, 4u};
        else { const u32x4 v = buf[u]; acc[0] += v[0]; acc[1] ^= v[1]; acc[2] += v[2]; acc[3] ^= v[3]; }
    }
    if (!STORE && acc[0] + acc[1] + acc[2] + acc[3] == 0x12345u) *sink = 1;
}

template <typename F> static float best_ms(F&& launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, ms = 0;
    for (int i = 0; i < 12; ++i) {
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= 3 && ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int npx = 64 * 2704;                                  // a batch-64 P3 map
    unsigned* sink; hipMalloc((void**)&sink, 4);
    u32x4* buf; hipMalloc((void**)&buf, (size_t)npx * 512);     // up to 256 channels
    hipMemset(buf, 1, (size_t)npx * 512);
    printf("%d pixels (batch-64 52x52 map); GB/s of the bytes really touched\n", npx);
    for (int ch : {32, 64, 128, 160, 256}) {
        const int pitch16 = ch * 2 / 16, pairs = ch / 32;
        const double bytes = (double)npx * ch * 2;
        for (int split : {1, pairs}) {
            if (split != 1 && pairs == 1) continue;
            const int waves = (npx + 15) / 16 * split, grid = (waves + 3) / 4;
            const float ts = best_ms([&] { hipLaunchKernelGGL(mfma_pattern<true>, dim3(grid), dim3(256), 0, 0, buf, npx, pitch16, pairs, split, sink); });
            const float tl = best_ms([&] { hipLaunchKernelGGL(mfma_pattern<false>, dim3(grid), dim3(256), 0, 0, buf, npx, pitch16, pairs, split, sink); });
            printf("  %3d channels (pixel pitch %3d B), mfma pattern, channel pairs of a pixel tile %s: store %6.0f GB/s (%5.1f us)   load %6.0f GB/s (%5.1f us)\n",
                   ch, ch * 2, split == 1 ? "in one wave     " : "split over waves", bytes / ts / 1e6, ts * 1e3, bytes / tl / 1e6, tl * 1e3);
        }
        const long units = (long)npx * pitch16;
        const float ts = best_ms([&] { hipLaunchKernelGGL(linear_pattern<true>, dim3(2048), dim3(256), 0, 0, buf, units, sink); });
        const float tl = best_ms([&] { hipLaunchKernelGGL(linear_pattern<false>, dim3(2048), dim3(256), 0, 0, buf, units, sink); });
        printf("  %3d channels, linear 16-byte units                                              : store %6.0f GB/s (%5.1f us)   load %6.0f GB/s (%5.1f us)\n",
               ch, bytes / ts / 1e6, ts * 1e3, bytes / tl / 1e6, tl * 1e3);
    }
    // a channel slice of a wider buffer (the concat buffers): 64 of 160 channels
    {
        const int pitch16 = 20, pairs = 2;
        const double bytes = (double)npx * 128;
        const int waves = (npx + 15) / 16, grid = (waves + 3) / 4;
        const float ts = best_ms([&] { hipLaunchKernelGGL(mfma_pattern<true>, dim3(grid), dim3(256), 0, 0, buf, npx, pitch16, pairs, 1, sink); });
        const float tl = best_ms([&] { hipLaunchKernelGGL(mfma_pattern<false>, dim3(grid), dim3(256), 0, 0, buf, npx, pitch16, pairs, 1, sink); });
        printf("   64 of 160 channels (pitch 320 B), mfma pattern: store %6.0f GB/s (%5.1f us)   load %6.0f GB/s (%5.1f us)\n", bytes / ts / 1e6, ts * 1e3, bytes / tl / 1e6, tl * 1e3);
    }
    return 0;
}
