// test_sharded_stub.cpp -- the lock-step multi-GPU caller of the gather library (host/zly_sharded.hpp) WITHOUT a GPU: ShardedDetectorT over a host-memory
// stand-in for the device runtime and link-time stubs of the two C ABIs it calls (zly_create / zly_destroy / zly_detect_device / zly_join / zly_slab_bytes
// / zly_default_config / zly_last_error; zly_gather_create / zly_gather_all / zly_gather_destroy / zly_gather_last_error).  TEST INFRASTRUCTURE: never linked
// into a product library; libzly.so has no CPU path.  The stub engine "detects" (slot + 1) % 4 boxes per frame whose fields encode (device, first pixel,
// frame tag); the stub gather is the all-gather's definition: every rank's send buffer, rank-major, into every rank's receive buffer.
//   test_sharded_stub <report.txt>
#include "../../zero-latency-yolo_amd/host/zly_sharded.hpp"

#include <fstream>
#include <map>

struct zly_engine { int device = 0, max_batch = 0, cap = 0; int calls = 0, joins = 0; };
struct zly_gather { std::vector<int> devices; int calls = 0; };
namespace {
int g_created = 0, g_destroyed = 0, g_gather_created = 0, g_gather_destroyed = 0;
std::string g_err;
struct HostDev {
    static bool setDevice(int) { return true; }
    static bool streamCreate(void** s) { *s = new int(0); return true; }
    static void streamDestroy(void* s) { delete (int*)s; }
    static bool streamSynchronize(void*) { return true; }
    static bool hostAlloc(void** p, size_t n) { *p = ::operator new(n); return true; }
    static void hostFree(void* p) { ::operator delete(p); }
    static bool deviceAlloc(void** p, size_t n) { *p = ::operator new(n); std::memset(*p, 0xEE, n); return true; }
    static void deviceFree(void* p) { ::operator delete(p); }
    static bool memsetAsync(void* p, int v, size_t n, void*) { std::memset(p, v, n); return true; }
    static bool copyH2DAsync(void* d, const void* h, size_t n, void*) { std::memcpy(d, h, n); return true; }
    static bool copyD2HAsync(void* h, const void* d, size_t n, void*) { std::memcpy(h, d, n); return true; }
};
}

extern "C" {
void zly_default_config(zly_config* c) { std::memset(c, 0, sizeof *c); c->model_w = 416; c->model_h = 416; c->conf_thr = 0.5f; c->iou_thr = 0.45f; c->max_batch = 1; c->max_dets = 64; c->warmup_runs = 3; c->use_graph = 1; }
const char* zly_last_error(void) { return g_err.c_str(); }
int32_t zly_create(const zly_config* cfg, zly_engine** out)
{
    if (std::string(cfg->weights_path) == "missing") { g_err = "stub: model not found"; return ZLY_ERR_MODEL_NOT_FOUND; }
    zly_engine* e = new zly_engine(); e->device = cfg->device; e->max_batch = cfg->max_batch; e->cap = cfg->max_dets; ++g_created; *out = e; return ZLY_OK;
}
int32_t zly_destroy(zly_engine* e) { ++g_destroyed; delete e; return ZLY_OK; }
size_t zly_slab_bytes(const zly_engine* e) { return sizeof(zly_slab_header) + (size_t)e->cap * sizeof(zly_det); }
int32_t zly_detect_device(zly_engine* e, int32_t n, const void* d_frames, int32_t w, int32_t h, void* d_slabs, uint32_t tag0, void*)
{
    if (n < 1 || n > e->max_batch) { g_err = "batch size out of range"; return ZLY_ERR_INVALID_ARGUMENT; }
    const size_t sb = zly_slab_bytes(e), fb = (size_t)w * h * 3;
    for (int i = 0; i < n; ++i) {
        unsigned char* slab = (unsigned char*)d_slabs + (size_t)i * sb;
        zly_slab_header hd{(i + 1) % 4, (i + 1) % 4 + 2, 0u, tag0 + (uint32_t)i};
        std::memcpy(slab, &hd, sizeof hd);
        for (int k = 0; k < hd.n_kept; ++k) {
            zly_det d; std::memset(&d, 0, sizeof d);
            d.x = (float)e->device; d.y = (float)((const uint8_t*)d_frames)[(size_t)i * fb]; d.w = (float)k; d.confidence = 0.9f; d.class_id = i;
            std::memcpy(slab + sizeof hd + (size_t)k * sizeof d, &d, sizeof d);
        }
    }
    e->calls++;
    return ZLY_OK;
}
int32_t zly_join(zly_engine* e, void*, int32_t) { e->joins++; return ZLY_OK; }
const char* zly_gather_last_error(void) { return g_err.c_str(); }
int32_t zly_gather_create(int32_t ndev, const int32_t* devices, zly_gather** out) { zly_gather* g = new zly_gather(); g->devices.assign(devices, devices + ndev); ++g_gather_created; *out = g; return 0; }
int32_t zly_gather_all(zly_gather* g, const void* const* d_send, void* const* d_recv, size_t bytes, void* const*)
{
    for (size_t i = 0; i < g->devices.size(); ++i)
        for (size_t r = 0; r < g->devices.size(); ++r) std::memcpy((unsigned char*)d_recv[i] + r * bytes, d_send[r], bytes);
    g->calls++;
    return 0;
}
int32_t zly_gather_ndev(const zly_gather* g) { return (int32_t)g->devices.size(); }
int32_t zly_gather_destroy(zly_gather* g) { ++g_gather_destroyed; delete g; return 0; }
}

using namespace zero_latency;

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    std::ofstream rep(argv[1]);
    ServerConfig config;
    config.model_path = "stub.zlyw";
    config.detection.model_width = 8; config.detection.model_height = 4;
    const size_t fb = 8 * 4 * 3;
    {
        ShardedDetectorT<HostDev> bad;
        ServerConfig c2 = config; c2.model_path = "missing";
        auto r = bad.initialize(c2, 2, 0, 3, 5);
        rep << "missing_model=" << (r.hasError() ? static_cast<int>(r.error().code) : 0) << "\n";
    }
    ShardedDetectorT<HostDev> det;
    if (det.initialize(config, 2, 4, 3, 5).hasError()) return 3;          // devices 4 and 5, three frames each per step
    rep << "devices=" << det.devices() << "\ncapacity=" << det.capacity() << "\n";
    auto make = [&](int n) {
        std::vector<InferenceRequest> reqs((size_t)n);
        for (int i = 0; i < n; ++i) {
            reqs[(size_t)i].frame_id = 500u + (uint32_t)i; reqs[(size_t)i].timestamp = 9000u + (uint64_t)i; reqs[(size_t)i].width = 8; reqs[(size_t)i].height = 4;
            reqs[(size_t)i].data.assign(fb, (uint8_t)(10 + i));
        }
        return reqs;
    };
    for (int n : {6, 5, 1}) {                                              // a full global batch, a ragged one, fewer frames than devices
        auto reqs = make(n);
        auto r = det.detectBatch(reqs);
        if (r.hasError()) { rep << "error_" << n << "=" << r.error().toString() << "\n"; return 4; }
        bool ok = r.value().size() == (size_t)n;
        for (int i = 0; i < n && ok; ++i) {
            const GameState& g = r.value()[(size_t)i];
            const int dev = 4 + i % 2, slot = i / 2;
            ok = g.frame_id == 500u + (uint32_t)i && g.timestamp == 9000u + (uint64_t)i && (int)g.detections.size() == (slot + 1) % 4;
            for (size_t k = 0; k < g.detections.size() && ok; ++k)
                ok = g.detections[k].box.x == (float)dev && g.detections[k].box.y == (float)(10 + i) && g.detections[k].box.width == (float)k && g.detections[k].class_id == slot;
        }
        rep << "batch_" << n << "_ok=" << ok << "\n";
    }
    InferenceRequest wrong; wrong.width = 8; wrong.height = 4; wrong.data.assign(fb - 1, 0);
    rep << "wrong_size=" << static_cast<int>(det.detectBatch({wrong}).error().code) << "\ntoo_many=" << static_cast<int>(det.detectBatch(make(7)).error().code) << "\n";
    det.shutdown();
    rep << "not_initialized=" << static_cast<int>(det.detectBatch(make(1)).error().code) << "\ncreated=" << g_created << "\ndestroyed=" << g_destroyed
        << "\ngathers_created=" << g_gather_created << "\ngathers_destroyed=" << g_gather_destroyed << "\n";
    return 0;
}
