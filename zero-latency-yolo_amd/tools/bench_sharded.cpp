// bench_sharded.cpp -- the lock-step multi-GPU form of the detect path in ONE process (host/zly_sharded.hpp): frames sharded one-per-GPU, result slabs
// all-gathered over RCCL/xGMI by libzly_gather.so, one download on device 0.  Host memory in, detections in host memory out.
//   zly_sharded_bench <weights.zlyw> <ndev> <seconds> <frames per device per step>
// First it checks the gathered result of one global batch against zly_detect_batch on a plain engine (same frames, same order: detection for detection);
// then it times steps.  Prints one JSON line.  On a one-GPU box ndev = 1 exercises the same code (RCCL communicator of one rank).
#include "zly_sharded_hip.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

using namespace zero_latency;
using Clock = std::chrono::steady_clock;

int main(int argc, char** argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: %s weights ndev seconds frames_per_device\n", argv[0]); return 2; }
    const int ndev = std::max(1, atoi(argv[2])), per = std::max(1, atoi(argv[4]));
    const double seconds = atof(argv[3]);
    ServerConfig config;
    config.model_path = argv[1];
    const int W = config.detection.model_width, H = config.detection.model_height;
    const size_t fb = (size_t)W * H * 3;
    const int B = ndev * per;
    std::vector<InferenceRequest> reqs((size_t)B);
    std::mt19937 rng(4242);
    for (int i = 0; i < B; ++i) {
        reqs[(size_t)i].client_id = (uint32_t)(i % 7); reqs[(size_t)i].frame_id = 1000u + (uint32_t)i; reqs[(size_t)i].timestamp = 77000ull + (uint64_t)i;
        reqs[(size_t)i].width = (uint16_t)W; reqs[(size_t)i].height = (uint16_t)H;
        reqs[(size_t)i].data.resize(fb);
        uint32_t* p = reinterpret_cast<uint32_t*>(reqs[(size_t)i].data.data());
        for (size_t k = 0; k < fb / 4; ++k) p[k] = rng();
    }
    ShardedDetector det;
    auto init = det.initialize(config, ndev, 0, per, 64);
    if (init.hasError()) { std::fprintf(stderr, "initialize: %s\n", init.error().toString().c_str()); return 3; }
    auto first = det.detectBatch(reqs);
    if (first.hasError()) { std::fprintf(stderr, "detectBatch: %s\n", first.error().toString().c_str()); return 4; }
    // reference: the same frames through zly_detect_batch on one plain engine
    int equal = 1;
    size_t dets_total = 0;
    {
        zly_config c; zly_default_config(&c);
        c.weights_path = argv[1]; c.max_batch = B; c.max_dets = 64; c.flags = ZLY_FLAG_NO_HEAD_TENSOR;
        zly_engine* e = nullptr;
        if (zly_create(&c, &e) != ZLY_OK) { std::fprintf(stderr, "zly_create: %s\n", zly_last_error()); return 3; }
        std::vector<const uint8_t*> ptrs((size_t)B); std::vector<size_t> nb((size_t)B, fb); std::vector<int32_t> ws((size_t)B, W), hs((size_t)B, H), n((size_t)B);
        for (int i = 0; i < B; ++i) ptrs[(size_t)i] = reqs[(size_t)i].data.data();
        std::vector<zly_det> out((size_t)B * 64);
        if (zly_detect_batch(e, B, ptrs.data(), nb.data(), ws.data(), hs.data(), out.data(), 64, n.data()) != ZLY_OK) { std::fprintf(stderr, "zly_detect_batch: %s\n", zly_last_error()); return 4; }
        for (int i = 0; i < B; ++i) {
            const GameState& g = first.value()[(size_t)i];
            const size_t k = (size_t)std::min(n[(size_t)i], 64);
            dets_total += k;
            if (g.frame_id != reqs[(size_t)i].frame_id || g.timestamp != reqs[(size_t)i].timestamp || g.detections.size() != k) { equal = 0; continue; }
            for (size_t j = 0; j < k; ++j) {                      // everything but the wall-clock timestamp of the detection
                const zly_det& a = out[(size_t)i * 64 + j]; const Detection& b = g.detections[j];
                if (a.x != b.box.x || a.y != b.box.y || a.w != b.box.width || a.h != b.box.height || a.confidence != b.confidence || a.class_id != b.class_id) equal = 0;
            }
        }
        zly_destroy(e);
    }
    // timed steps
    for (int i = 0; i < 5; ++i) (void)det.detectBatch(reqs);
    const auto t0 = Clock::now();
    size_t steps = 0;
    while (std::chrono::duration<double>(Clock::now() - t0).count() < seconds) { if (det.detectBatch(reqs).hasError()) return 5; ++steps; }
    const double dt = std::chrono::duration<double>(Clock::now() - t0).count();
    std::printf("{\"mode\":\"sharded\",\"devices\":%d,\"rccl_ranks\":%d,\"frames_per_device_per_step\":%d,\"steps\":%zu,\"frames_per_sec\":%.1f,\"ms_per_step\":%.4f,"
                "\"gathered_bytes_per_step\":%zu,\"equals_zly_detect_batch\":%d,\"detections_first_batch\":%zu}\n",
                det.devices(), det.devices(), per, steps, (double)steps * B / dt, dt / (double)steps * 1e3, det.gatheredBytesPerStep(), equal, dets_total);
    det.shutdown();
    return equal ? 0 : 6;
}
