// zly_sharded_hip.hpp -- the HIP runtime behind ShardedDetectorT (host/zly_sharded.hpp): device selection, streams, pinned / device memory, async copies.
// Plumbing only: every kernel runs inside libzly.so, the collective inside libzly_gather.so.
#pragma once
#include "zly_sharded.hpp"
#include <hip/hip_runtime_api.h>

namespace zero_latency {
struct HipDev {
    static bool setDevice(int d) { return hipSetDevice(d) == hipSuccess; }
    static bool streamCreate(void** s) { hipStream_t st = nullptr; const bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess; *s = st; return ok; }
    static void streamDestroy(void* s) { (void)hipStreamDestroy((hipStream_t)s); }
    static bool streamSynchronize(void* s) { return hipStreamSynchronize((hipStream_t)s) == hipSuccess; }
    static bool hostAlloc(void** p, size_t n) { return hipHostMalloc(p, n, hipHostMallocDefault) == hipSuccess; }
    static void hostFree(void* p) { (void)hipHostFree(p); }
    static bool deviceAlloc(void** p, size_t n) { return hipMalloc(p, n) == hipSuccess; }
    static void deviceFree(void* p) { (void)hipFree(p); }
    static bool memsetAsync(void* p, int v, size_t n, void* s) { return hipMemsetAsync(p, v, n, (hipStream_t)s) == hipSuccess; }
    static bool copyH2DAsync(void* d, const void* h, size_t n, void* s) { return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s) == hipSuccess; }
    static bool copyD2HAsync(void* h, const void* d, size_t n, void* s) { return hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)s) == hipSuccess; }
};
using ShardedDetector = ShardedDetectorT<HipDev>;
}  // namespace zero_latency
