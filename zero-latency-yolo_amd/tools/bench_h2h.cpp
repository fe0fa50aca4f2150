// bench_h2h.cpp -- host-to-host throughput of the detect path ("request bytes in host memory ->
// detections in host memory", config 3: >= 8 submitting host threads), measured natively so that no interpreter sits
// between the threads and the C ABI.  bench.py runs it as a child process and embeds its JSON line.
//
//   zly_h2h_bench <weights.zlyw> <cabi|plugin|lone> <threads> <seconds> <max_batch> [engines [w h]]
//
// engines > 1: that many engine instances on the GPU (ZLY_FLAG_SINGLE_CHAIN), submitting thread t feeds engine t % engines; their
// batches overlap on the device (bench.py --engines)
//
// cabi  : T threads call zly_submit (one copy of the frame into the engine's pinned ring, on the calling thread), one
//         consumer thread calls zly_wait in ticket order -- the shape of the reference's submitInference / result hand-over
//         (onnx_engine.cpp:223-261, 355-364).
// plugin: the same load through HipInferenceEngine::submitInference / InferenceCallback, i.e. what the reference's
//         NetworkServer would drive (network_server.cpp:184-224, 243-283).
// lone  : the latency a single client of the server sees: ONE thread, HipInferenceEngine::submitInference of one frame, wait for its
//         InferenceCallback, submit the next (reference: one client's frame through runInference, onnx_engine.cpp:518-646); p50 / p99 of
//         submit -> callback.  The pipelined path serves a lone frame at once (no batching window) by replaying the batch-1 graph.
// Frames are u8 BGR noise in ordinary (pageable) host memory, 4 distinct frames per thread.
#include "zly_compat.hpp"
#include "hip_inference_engine.h"
#include "zly.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <random>
#include <thread>

using namespace zero_latency;
using Clock = std::chrono::steady_clock;

static double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct Lat {
    std::vector<float> ms;
    void add(double v) { if (ms.size() < (1u << 22)) ms.push_back((float)v); }
    double pct(double p) { if (ms.empty()) return 0; std::sort(ms.begin(), ms.end()); return ms[std::min(ms.size() - 1, (size_t)(p * ms.size()))]; }
};

int main(int argc, char** argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: %s weights cabi|plugin threads seconds max_batch [w h]\n", argv[0]); return 2; }
    const std::string weights = argv[1], mode = argv[2];
    const int T = std::max(1, atoi(argv[3]));
    const double seconds = atof(argv[4]);
    const int max_batch = std::max(1, atoi(argv[5]));
    const int E = argc > 6 ? std::max(1, atoi(argv[6])) : 1;
    const int W = argc > 8 ? atoi(argv[7]) : 416, H = argc > 8 ? atoi(argv[8]) : 416;
    const size_t fb = (size_t)W * H * 3;

    std::vector<std::vector<std::vector<uint8_t>>> frames((size_t)T);
    for (int t = 0; t < T; ++t) {
        std::mt19937 rng(1234u + (unsigned)t);
        frames[(size_t)t].resize(4);
        for (auto& f : frames[(size_t)t]) {
            f.resize(fb);
            uint32_t* p = reinterpret_cast<uint32_t*>(f.data());
            for (size_t i = 0; i < fb / 4; ++i) p[i] = rng();
        }
    }

    std::atomic<bool> go{false}, stop{false};
    std::atomic<uint64_t> submitted{0}, completed{0}, dets_total{0}, errors{0};
    Lat lat;
    std::mutex lat_mu;
    Clock::time_point t0, t1;
    uint64_t warm_completed = 0;

    if (mode == "cabi") {
        struct Item { uint64_t ticket; Clock::time_point ts; };
        struct Eng {
            zly_engine* e = nullptr;
            std::mutex qmu;
            std::condition_variable qcv;
            std::deque<Item> q;
            std::atomic<uint64_t> submitted{0}, done{0};
        };
        std::vector<std::unique_ptr<Eng>> engs;
        for (int i = 0; i < E; ++i) {
            zly_config c;
            zly_default_config(&c);
            c.weights_path = weights.c_str();
            c.model_w = 416; c.model_h = 416;
            c.max_batch = max_batch; c.max_dets = 64; c.warmup_runs = 3;
            c.flags = ZLY_FLAG_NO_HEAD_TENSOR | (E > 1 ? ZLY_FLAG_SINGLE_CHAIN : ZLY_FLAG_ASYNC_NMS);
            engs.emplace_back(new Eng());
            if (zly_create(&c, &engs.back()->e) != ZLY_OK) { std::fprintf(stderr, "zly_create: %s\n", zly_last_error()); return 3; }
        }
        std::vector<std::thread> subs, cons;
        for (int t = 0; t < T; ++t)
            subs.emplace_back([&, t] {
                Eng& g = *engs[(size_t)(t % E)];
                while (!go) std::this_thread::yield();
                size_t k = 0;
                while (!stop) {
                    const auto& f = frames[(size_t)t][k++ & 3];
                    uint64_t ticket = 0;
                    const auto ts = Clock::now();
                    if (zly_submit(g.e, f.data(), f.size(), W, H, &ticket) != ZLY_OK) { if (errors++ == 0) std::fprintf(stderr, "zly_submit: %s\n", zly_last_error()); break; }
                    submitted++; g.submitted++;
                    { std::lock_guard<std::mutex> lk(g.qmu); g.q.push_back(Item{ticket, ts}); }
                    g.qcv.notify_one();
                }
            });
        for (int i = 0; i < E; ++i)
            cons.emplace_back([&, i] {
                Eng& g = *engs[(size_t)i];
                std::vector<zly_det> dets(64);
                while (true) {
                    Item it;
                    {
                        std::unique_lock<std::mutex> lk(g.qmu);
                        g.qcv.wait(lk, [&] { return !g.q.empty() || (stop && g.submitted == g.done); });
                        if (g.q.empty()) return;
                        it = g.q.front(); g.q.pop_front();
                    }
                    int32_t n = 0;
                    const int32_t rc = zly_wait(g.e, it.ticket, dets.data(), 64, &n);
                    g.done++;
                    if (rc != ZLY_OK) { if (errors++ == 0) std::fprintf(stderr, "zly_wait: %d %s\n", rc, zly_last_error()); continue; }
                    const double ms = secs(it.ts, Clock::now()) * 1e3;
                    dets_total += (uint64_t)std::min(n, 64);
                    completed++;
                    std::lock_guard<std::mutex> lk(lat_mu);
                    lat.add(ms);
                }
            });
        go = true;
        std::this_thread::sleep_for(std::chrono::milliseconds(500));                 // warm-up: graphs captured, rings in steady state
        { std::lock_guard<std::mutex> lk(lat_mu); lat.ms.clear(); }
        warm_completed = completed; t0 = Clock::now();
        std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
        const uint64_t end_completed = completed; t1 = Clock::now();
        stop = true;
        for (auto& th : subs) th.join();
        for (auto& g : engs) g->qcv.notify_all();
        for (auto& th : cons) th.join();
        zly_stats st{};
        uint64_t batches = 0, count = 0, sampled = 0;
        double pre = 0, fwd = 0, post = 0;
        for (auto& g : engs) {
            zly_get_stats(g->e, &st);
            batches += st.batches; count += st.inference_count; sampled += st.sampled_frames;
            pre += st.sampled_preprocess_ms; fwd += st.sampled_forward_ms; post += st.sampled_postprocess_ms;
        }
        const double dt = secs(t0, t1);
        const double fps = (double)(end_completed - warm_completed) / dt;
        {
            std::lock_guard<std::mutex> lk(lat_mu);
            std::printf("{\"mode\":\"cabi\",\"threads\":%d,\"engines\":%d,\"max_batch\":%d,\"frame\":\"%dx%d\",\"seconds\":%.3f,\"frames\":%llu,\"frames_per_sec\":%.1f,"
                        "\"pcie_h2d_GBps\":%.2f,\"avg_batch\":%.1f,\"p50_ms\":%.3f,\"p99_ms\":%.3f,\"errors\":%llu,\"detections\":%llu,"
                        "\"avg_preprocess_ms_per_frame\":%.5f,\"avg_forward_ms_per_frame\":%.5f,\"avg_postprocess_ms_per_frame\":%.5f}\n",
                        T, E, max_batch, W, H, dt, (unsigned long long)(end_completed - warm_completed), fps, fps * (double)fb / 1e9,
                        batches ? (double)count / (double)batches : 0.0, lat.pct(0.5), lat.pct(0.99),
                        (unsigned long long)errors.load(), (unsigned long long)dets_total.load(),
                        sampled ? pre / (double)sampled : 0.0, sampled ? fwd / (double)sampled : 0.0, sampled ? post / (double)sampled : 0.0);
        }
        for (auto& g : engs) zly_destroy(g->e);
        return errors ? 4 : 0;
    }

    // ---- plugin mode -------------------------------------------------------------------------------------------------
    setenv("ZLY_MAX_BATCH", std::to_string(max_batch).c_str(), 1);
    setenv("ZLY_ENGINES_PER_GPU", std::to_string(E).c_str(), 1);
    setenv("ZLY_MAX_DETS", "64", 1);
    setenv("ZLY_MODEL_WATCH_MS", "0", 1);
    ServerConfig config;
    config.model_path = weights;
    config.inference_engine = "hip";
    std::unique_ptr<IInferenceEngine> engine = InferenceEngineManager::getInstance().createEngine("hip", config);
    if (!engine) { std::fprintf(stderr, "factory 'hip' not registered\n"); return 3; }
    engine->setCallback([&](uint32_t, const GameState& st) {
        // GameState.timestamp echoes the request's: the submitter put its steady-clock microseconds there
        const uint64_t now_us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(Clock::now().time_since_epoch()).count();
        dets_total += st.detections.size();
        completed++;
        std::lock_guard<std::mutex> lk(lat_mu);
        lat.add((double)(now_us - st.timestamp) / 1e3);
    });
    auto init = engine->initialize();
    if (init.hasError()) { std::fprintf(stderr, "initialize: %s\n", init.error().toString().c_str()); return 3; }
    if (mode == "lone") {
        InferenceRequest r;
        r.client_id = 7; r.width = (uint16_t)W; r.height = (uint16_t)H;
        Lat sub;                                                     // time inside submitInference itself (the copy into the pinned ring)
        const auto run_until = [&](double secs_) {
            const auto end = Clock::now() + std::chrono::duration<double>(secs_);
            uint32_t k = 0;
            while (Clock::now() < end) {
                r.data = frames[0][k & 3];
                r.frame_id = k++;
                const uint64_t want = completed.load() + 1;
                const auto ts = Clock::now();
                r.timestamp = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(ts.time_since_epoch()).count();
                if (engine->submitInference(r).hasError()) { errors++; return; }
                sub.add(secs(ts, Clock::now()) * 1e3);
                while (completed.load() < want) { /* spin: the client's own receive loop is not what is measured */ }
            }
        };
        run_until(0.3);
        { std::lock_guard<std::mutex> lk(lat_mu); lat.ms.clear(); }
        sub.ms.clear();
        t0 = Clock::now();
        run_until(seconds);
        t1 = Clock::now();
        auto status = engine->getStatus();
        engine->shutdown();
        std::lock_guard<std::mutex> lk(lat_mu);
        std::printf("{\"mode\":\"lone\",\"engines\":%d,\"max_batch\":%d,\"frame\":\"%dx%d\",\"seconds\":%.3f,\"frames\":%zu,\"frames_per_sec\":%.1f,"
                    "\"p50_ms\":%.4f,\"p90_ms\":%.4f,\"p99_ms\":%.4f,\"submit_p50_ms\":%.4f,\"errors\":%llu,\"batches\":%s}\n",
                    E, max_batch, W, H, secs(t0, t1), lat.ms.size(), (double)lat.ms.size() / secs(t0, t1), lat.pct(0.5), lat.pct(0.9), lat.pct(0.99), sub.pct(0.5),
                    (unsigned long long)errors.load(), status["batches"].c_str());
        return errors ? 4 : 0;
    }
    std::vector<std::thread> subs;
    for (int t = 0; t < T; ++t)
        subs.emplace_back([&, t] {
            std::vector<InferenceRequest> reqs(4);
            for (int k = 0; k < 4; ++k) {
                reqs[(size_t)k].client_id = (uint32_t)t; reqs[(size_t)k].width = (uint16_t)W; reqs[(size_t)k].height = (uint16_t)H;
                reqs[(size_t)k].data = frames[(size_t)t][(size_t)k];
            }
            while (!go) std::this_thread::yield();
            uint32_t k = 0;
            while (!stop) {
                InferenceRequest& r = reqs[k & 3];
                r.frame_id = k++;
                r.timestamp = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(Clock::now().time_since_epoch()).count();
                if (engine->submitInference(r).hasError()) { errors++; break; }
                submitted++;
            }
        });
    go = true;
    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    { std::lock_guard<std::mutex> lk(lat_mu); lat.ms.clear(); }
    warm_completed = completed; t0 = Clock::now();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    const uint64_t end_completed = completed; t1 = Clock::now();
    stop = true;
    for (auto& th : subs) th.join();
    auto status = engine->getStatus();
    engine->shutdown();
    const double dt = secs(t0, t1);
    const double fps = (double)(end_completed - warm_completed) / dt;
    std::lock_guard<std::mutex> lk(lat_mu);
    std::printf("{\"mode\":\"plugin\",\"threads\":%d,\"engines\":%d,\"max_batch\":%d,\"frame\":\"%dx%d\",\"seconds\":%.3f,\"frames\":%llu,\"frames_per_sec\":%.1f,"
                "\"pcie_h2d_GBps\":%.2f,\"p50_ms\":%.3f,\"p99_ms\":%.3f,\"errors\":%llu,\"detections\":%llu,\"batches\":%s,"
                "\"avg_preprocessing_time_ms\":%s,\"avg_postprocessing_time_ms\":%s}\n",
                T, E, max_batch, W, H, dt, (unsigned long long)(end_completed - warm_completed), fps, fps * (double)fb / 1e9, lat.pct(0.5), lat.pct(0.99),
                (unsigned long long)errors.load(), (unsigned long long)dets_total.load(), status["batches"].c_str(),
                status["avg_preprocessing_time_ms"].c_str(), status["avg_postprocessing_time_ms"].c_str());
    return errors ? 4 : 0;
}
