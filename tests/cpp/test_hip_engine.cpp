// test_hip_engine.cpp -- drives the MI355X engine exactly as the reference's server does: through
// InferenceEngineManager / IInferenceEngine (reference src/server/main.cpp:224-246,
// src/network/network_server.cpp:21-22,200-209,243-283), never through the C ABI directly.
//
//   test_hip_engine <weights.zlyw> <frames.bin> <out.json> [probe | reload <new_weights.zlyw>]
//
// reload: after the first pass the model file is replaced by <new_weights.zlyw> (rename over it); the driver waits for the
// engine's watcher (ZLY_MODEL_WATCH_MS) to pick it up (status model_version 2), submits the same frames again and reports
// both passes ("results", "results_after_reload") and both status maps.
//
// frames.bin: u32 count, then per frame {u16 width, u16 height, u32 nbytes, bytes[nbytes]} (a frame whose
// nbytes != w*h*3 exercises the INVALID_INPUT path).  Writes one JSON document with the detections
// every callback delivered, the error behaviour observed, and getStatus(); tests/test_host_engine.py
// compares it with the ctypes path and the CPU oracle.
#include "zly_compat.hpp"
#include "hip_inference_engine.h"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>

using namespace zero_latency;

struct Frame { uint16_t w, h; std::vector<uint8_t> data; };

static bool readFrames(const char* path, std::vector<Frame>* out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    uint32_t n = 0;
    f.read(reinterpret_cast<char*>(&n), 4);
    for (uint32_t i = 0; i < n; ++i) {
        Frame fr;
        uint32_t nb = 0;
        f.read(reinterpret_cast<char*>(&fr.w), 2);
        f.read(reinterpret_cast<char*>(&fr.h), 2);
        f.read(reinterpret_cast<char*>(&nb), 4);
        fr.data.resize(nb);
        f.read(reinterpret_cast<char*>(fr.data.data()), nb);
        if (!f) return false;
        out->push_back(std::move(fr));
    }
    return true;
}

int main(int argc, char** argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: %s weights frames.bin out.json\n", argv[0]); return 2; }
    std::vector<Frame> frames;
    if (!readFrames(argv[2], &frames)) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }

    ServerConfig config;                          // defaults = configs/server.json (conf 0.5, nms 0.45, 416x416)
    config.model_path = argv[1];
    config.inference_engine = "hip";

    std::ostringstream js;
    js << "{";

    // engine selection by name, as server/main.cpp:226-240 does for any name other than "onnx"
    auto& mgr = InferenceEngineManager::getInstance();
    js << "\"available\":" << (mgr.isEngineAvailable("hip") ? "true" : "false") << ",";
    std::unique_ptr<IInferenceEngine> engine = mgr.createEngine(config.inference_engine, config);
    if (!engine) { std::fprintf(stderr, "factory 'hip' not registered\n"); return 3; }
    js << "\"name\":\"" << engine->getName() << "\",";

    // submit before initialize -> NOT_INITIALIZED (onnx_engine.cpp:224-226)
    InferenceRequest probe;
    probe.width = 4; probe.height = 4; probe.data.assign(48, 0);
    js << "\"submit_before_init\":" << static_cast<int>(engine->submitInference(probe).error().code) << ",";

    // a missing model is an error from initialize(), never random "simulation" boxes
    {
        ServerConfig bad = config;
        bad.model_path = "/nonexistent/model.zlyw";
        auto e2 = mgr.createEngine("hip", bad);
        js << "\"init_missing_model\":" << static_cast<int>(e2->initialize().error().code) << ",";
    }

    std::mutex mu;
    std::condition_variable cv;
    struct Got { uint32_t client_id, frame_id; uint64_t timestamp; std::vector<Detection> dets; };
    std::vector<Got> got;
    engine->setCallback([&](uint32_t client_id, const GameState& st) {
        std::lock_guard<std::mutex> lk(mu);
        got.push_back(Got{client_id, st.frame_id, st.timestamp, st.detections});
        cv.notify_all();
    });

    auto init = engine->initialize();
    js << "\"init\":" << static_cast<int>(init.error().code) << ",";
    const bool probe_only = argc > 4 && std::string(argv[4]) == "probe";      // CPU-only check of the plugin plumbing
    if (init.hasError() || probe_only) {
        if (init.isOk()) engine->shutdown();
        js << "\"probe\":true}";
        std::ofstream(argv[3]) << js.str() << "\n";
        if (init.hasError()) std::fprintf(stderr, "initialize failed: %s\n", init.error().toString().c_str());
        return probe_only ? 0 : 4;
    }

    // ZLY_TEST_POLL_STATUS=1: a second thread calls getStatus() / getQueueSize() in a tight loop while the burst is served, as the
    // reference's monitor thread does every 5 s (server/main.cpp:103-104): no call may wait for a running batch
    std::atomic<bool> poll_stop{false};
    std::atomic<long> poll_calls{0};
    double poll_max_ms = 0;
    std::thread poller;
    if (std::getenv("ZLY_TEST_POLL_STATUS")) {
        poller = std::thread([&] {
            while (!poll_stop) {
                const auto t0 = std::chrono::steady_clock::now();
                auto st = engine->getStatus();
                (void)engine->getQueueSize();
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (ms > poll_max_ms) poll_max_ms = ms;
                poll_calls++;
            }
        });
    }
    size_t expected = 0;
    for (size_t i = 0; i < frames.size(); ++i) {
        InferenceRequest r;
        r.client_id = 1000 + (uint32_t)(i % 3);
        r.frame_id = (uint32_t)i;
        r.timestamp = 777000 + i;
        r.width = frames[i].w; r.height = frames[i].h;
        r.data = frames[i].data;
        r.is_keyframe = (i % 10) == 0;
        if (r.data.size() == (size_t)r.width * r.height * 3u) ++expected;
        auto res = engine->submitInference(r);
        if (res.hasError()) { std::fprintf(stderr, "submit failed: %s\n", res.error().toString().c_str()); return 5; }
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        if (!cv.wait_for(lk, std::chrono::seconds(60), [&] { return got.size() >= expected; })) {
            std::fprintf(stderr, "timeout: %zu of %zu callbacks\n", got.size(), expected);
            return 6;
        }
    }
    // give a wrongly delivered extra callback (for the invalid frame) a moment to show up
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    if (poller.joinable()) {
        poll_stop = true;
        poller.join();
        js << "\"status_poll\":{\"calls\":" << poll_calls.load() << ",\"max_ms\":" << poll_max_ms << "},";
    }

    auto status = engine->getStatus();
    js << "\"queue_size_after\":" << engine->getQueueSize() << ",";
    js << "\"status\":{";
    bool first = true;
    for (const auto& kv : status) { js << (first ? "" : ",") << "\"" << kv.first << "\":\"" << kv.second << "\""; first = false; }
    js << "},\"results\":[";
    {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < got.size(); ++i) {
            const Got& g = got[i];
            js << (i ? "," : "") << "{\"client_id\":" << g.client_id << ",\"frame_id\":" << g.frame_id << ",\"timestamp\":" << g.timestamp << ",\"dets\":[";
            for (size_t k = 0; k < g.dets.size(); ++k) {
                const Detection& d = g.dets[k];
                uint32_t bits[5];
                std::memcpy(bits, &d, 20);                      // exact float bits: box + confidence
                js << (k ? "," : "") << "[" << bits[0] << "," << bits[1] << "," << bits[2] << "," << bits[3] << "," << bits[4] << ","
                   << d.class_id << "," << d.track_id << "," << d.timestamp << "]";
            }
            js << "]}";
        }
    }
    js << "],";
    if (argc > 5 && std::string(argv[4]) == "reload") {
        const std::string hash0 = status["model_hash"];
        if (std::rename(argv[5], argv[1]) != 0) { std::fprintf(stderr, "rename failed\n"); return 7; }
        bool swapped = false;
        for (int i = 0; i < 600 && !swapped; ++i) {                   // up to 30 s
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
            swapped = engine->getStatus()["model_version"] == "2";
        }
        if (!swapped) { std::fprintf(stderr, "watcher did not reload\n"); return 8; }
        size_t base;
        { std::lock_guard<std::mutex> lk(mu); base = got.size(); }
        for (size_t i = 0; i < frames.size(); ++i) {
            InferenceRequest r;
            r.client_id = 2000; r.frame_id = 100 + (uint32_t)i; r.timestamp = 888000 + i;
            r.width = frames[i].w; r.height = frames[i].h; r.data = frames[i].data;
            if (engine->submitInference(r).hasError()) return 5;
        }
        {
            std::unique_lock<std::mutex> lk(mu);
            if (!cv.wait_for(lk, std::chrono::seconds(60), [&] { return got.size() >= base + expected; })) return 6;
        }
        auto st2 = engine->getStatus();
        js << "\"hash_before\":\"" << hash0 << "\",\"status_after_reload\":{";
        bool f2 = true;
        for (const auto& kv : st2) { js << (f2 ? "" : ",") << "\"" << kv.first << "\":\"" << kv.second << "\""; f2 = false; }
        js << "},\"results_after_reload\":[";
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = base; i < got.size(); ++i) {
            const Got& g = got[i];
            js << (i > base ? "," : "") << "{\"client_id\":" << g.client_id << ",\"frame_id\":" << g.frame_id << ",\"timestamp\":" << g.timestamp << ",\"dets\":[";
            for (size_t k = 0; k < g.dets.size(); ++k) {
                const Detection& d = g.dets[k];
                uint32_t bits[5];
                std::memcpy(bits, &d, 20);
                js << (k ? "," : "") << "[" << bits[0] << "," << bits[1] << "," << bits[2] << "," << bits[3] << "," << bits[4] << ","
                   << d.class_id << "," << d.track_id << "," << d.timestamp << "]";
            }
            js << "]}";
        }
        js << "],";
        // a file that does not load: the old model keeps serving, version unchanged, reloadModel() reports the error
        { std::ofstream bad(argv[1], std::ios::binary | std::ios::trunc); bad << "not a model"; }
        auto* hip = dynamic_cast<HipInferenceEngine*>(engine.get());
        js << "\"reload_bad_file\":" << (hip ? static_cast<int>(hip->reloadModel().error().code) : -1) << ",";
        js << "\"version_after_bad_file\":\"" << engine->getStatus()["model_version"] << "\",";
    }
    auto sd = engine->shutdown();
    js << "\"shutdown_ok\":" << (sd.isOk() ? "true" : "false") << ",";
    js << "\"submit_after_shutdown\":" << static_cast<int>(engine->submitInference(probe).error().code) << "}";

    std::ofstream out(argv[3]);
    out << js.str() << "\n";
    return 0;
}
