// gather.cpp -- libzly_gather.so: the in-process RCCL all-gather of result slabs (include/zly_gather.h).  Host code + RCCL only.
#include "zly_gather.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string>
#include <vector>

static thread_local std::string g_gather_error;
static int gfail(int code, const std::string& msg) { g_gather_error = msg; return code; }

struct zly_gather {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};

extern "C" {

const char* zly_gather_last_error(void) { return g_gather_error.c_str(); }

int32_t zly_gather_create(int32_t ndev, const int32_t* devices, zly_gather** out)
{
    if (!out || !devices || ndev < 1 || ndev > 64) return gfail(2, "bad argument");
    *out = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return gfail(300, "no HIP device available");
    for (int i = 0; i < ndev; ++i) {
        if (devices[i] < 0 || devices[i] >= have) return gfail(2, "device ordinal out of range");
        for (int j = 0; j < i; ++j) if (devices[j] == devices[i]) return gfail(2, "a device may take part once");
    }
    zly_gather* g = new zly_gather();
    g->devices.assign(devices, devices + ndev);
    g->comms.resize((size_t)ndev);
    const ncclResult_t r = ncclCommInitAll(g->comms.data(), ndev, g->devices.data());
    if (r != ncclSuccess) { const std::string m = std::string("ncclCommInitAll: ") + ncclGetErrorString(r); delete g; return gfail(300, m); }
    *out = g;
    return 0;
}

int32_t zly_gather_ndev(const zly_gather* g) { return g ? (int32_t)g->devices.size() : 0; }

int32_t zly_gather_all(zly_gather* g, const void* const* d_send, void* const* d_recv, size_t bytes_per_rank, void* const* streams)
{
    if (!g || !d_send || !d_recv || !streams || bytes_per_rank == 0) return gfail(2, "bad argument");
    const size_t n = g->devices.size();
    for (size_t i = 0; i < n; ++i) if (!d_send[i] || !d_recv[i]) return gfail(2, "null buffer");
    ncclResult_t r = ncclGroupStart();
    for (size_t i = 0; i < n && r == ncclSuccess; ++i)
        r = ncclAllGather(d_send[i], d_recv[i], bytes_per_rank, ncclUint8, g->comms[i], (hipStream_t)streams[i]);
    const ncclResult_t r2 = ncclGroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return gfail(300, std::string("ncclAllGather: ") + ncclGetErrorString(r));
    return 0;
}

int32_t zly_gather_destroy(zly_gather* g)
{
    if (!g) return 0;
    for (ncclComm_t c : g->comms) if (c) ncclCommDestroy(c);
    delete g;
    return 0;
}

}  // extern "C"
