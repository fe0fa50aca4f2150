// hip_inference_engine.h -- the reference's inference-engine plugin, MI355X-native.
//
// `HipInferenceEngine` implements zero_latency::IInferenceEngine (reference
// src/inference/inference_engine.h:33-43) and `HipInferenceEngineFactory` registers it under the
// name "hip", so the reference's server selects it with `"inference_engine": "hip"` in
// configs/server.json (reference src/server/main.cpp:224-240).  It is the counterpart of
// OnnxInferenceEngine (reference src/inference/onnx_engine.{h,cpp}) and only talks to the GPU
// through the C ABI of include/zly.h.
//
// Behaviour kept from the reference:
//   * submitInference returns NOT_INITIALIZED when the engine is not running (onnx_engine.cpp:224-226),
//     copies the request and never blocks on inference (:233-258);
//   * GameState.frame_id / timestamp echo the request (:520-521); the callback runs on an engine-owned
//     thread, once per successful frame, and is not invoked for failed frames (:376-388);
//   * getStatus() exposes the reference's keys (:279-312);
//   * the model file is watched (SHA-256 every 10 s, :473-515; ZLY_MODEL_WATCH_MS overrides, 0 = off) and reloaded when it
//     changes: new engines are built beside the running ones and swapped in between two batches, so no request is
//     dropped or served by a half-loaded model; a file that fails to load leaves the old model serving (as the reference).
// Deliberate differences (DESIGN.md "host side"):
//   * every successful frame reaches the callback, in submission order (the reference drops frames
//     popped by its worker threads, :459-460);
//   * submitInference copies the request's pixels ONCE, straight into the engine's pinned staging ring, on the
//     caller's thread (zly_submit); the engine batches whatever is pending -- no batching window, so no added latency --
//     and overlaps upload, compute and download of consecutive batches (the reference's "dynamic batching" is a TODO
//     that runs frames one by one, :348-365).  One completion thread hands results to the callback (zly_wait);
//   * no SILENT simulation mode: a missing/bad model file is an error from initialize(), not random boxes (:70-75,105-110).
//     The reference's fake backend exists only as an explicit opt-in, ZLY_SIMULATE=1 ("keep only as an explicit --simulate
//     fallback"): initialize() then creates no engine and every frame gets generateRandomDetections' 0-5 random boxes (:1133-1177) through
//     the same ordered callback path; getStatus() says simulation_mode = true.  Nothing falls back to it on its own;
//   * one engine per GPU (ZLY_NUM_DEVICES, default 1; requests go round robin) instead of CPU worker threads.
#pragma once

#include "zly_compat.hpp"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

struct zly_engine;

namespace zero_latency {

class HipInferenceEngine : public IInferenceEngine {
public:
    explicit HipInferenceEngine(const ServerConfig& config);
    ~HipInferenceEngine() override;

    Result<void> initialize() override;
    Result<void> shutdown() override;
    Result<void> submitInference(const InferenceRequest& request) override;
    void setCallback(InferenceCallback callback) override;
    size_t getQueueSize() const override;
    std::string getName() const override;
    std::unordered_map<std::string, std::string> getStatus() const override;

    // hot reload (reference loadModel(path, true) from modelMonitorThreadFunc): also callable directly
    Result<void> reloadModel();

private:
    struct EngineHandle;                                   // owns a zly_engine*; destroyed with its last pending request
    struct Pending {
        uint64_t seq = 0;
        std::shared_ptr<EngineHandle> engine;
        uint64_t ticket = 0;
        bool failed = false;                               // refused by zly_submit (wrong byte count): no callback, keeps the sequence dense
        bool simulated = false;                            // ZLY_SIMULATE=1: no engine, no ticket; the completion thread draws the reference's random boxes
        uint32_t client_id = 0, frame_id = 0;
        uint64_t timestamp = 0, enqueue_ms = 0;
        size_t slot = 0;                                   // index of its queue in pending_
    };

    void completionLoop();
    void monitorLoop();
    void reaperLoop();
    void retire(std::shared_ptr<EngineHandle>&& h);         // drop a reference on the reaper thread (the last one destroys the engine there)
    void release(std::shared_ptr<EngineHandle>&& h);        // a request is done with its engine: plain drop while the engine serves, retire() once a reload replaced it
    std::vector<Detection> generateRandomDetections();      // onnx_engine.cpp:1133-1177 (ZLY_SIMULATE=1 only)
    std::shared_ptr<EngineHandle> createEngineOn(int device, int32_t* rc, std::string* msg) const;

    ServerConfig config_;
    int max_batch_ = 64;
    int max_dets_ = 256;
    mutable std::mutex engines_mutex_;                     // guards the vector (held for pointer copies only, never across a device call)
    std::vector<std::shared_ptr<EngineHandle>> engines_;   // one per GPU and engine instance
    // what submitInference reads: an immutable snapshot of engines_, swapped as a whole by initialize / reloadModel / shutdown (std::atomic_load /
    // atomic_store on the shared_ptr) -- twelve submitting threads used to take engines_mutex_ for every frame
    std::shared_ptr<const std::vector<std::shared_ptr<EngineHandle>>> engines_snapshot_;
    bool simulate_ = false;                                // ZLY_SIMULATE=1 (explicit opt-in, see above)
    uint32_t sim_rng_state_[4] = {0, 0, 0, 0};             // completion thread only
    int first_device_ = 0;
    int engines_per_gpu_ = 1;
    std::thread monitor_, completer_, reaper_;
    // Engines replaced by a hot reload are destroyed on the reaper thread: zly_destroy drains the engine's streams and takes the process-wide
    // exclusive gate, which must never stall callback delivery (the completion thread) or a submitting thread.
    std::mutex reap_mutex_;
    std::condition_variable reap_cv_;
    std::vector<std::shared_ptr<EngineHandle>> retired_;
    bool reap_stop_ = false;
    std::mutex reload_mutex_;                              // one reload at a time
    std::atomic<uint32_t> model_version_{1};
    std::string model_hash_;                               // guarded by stats_mutex_
    std::atomic<bool> running_{false};

    struct Done { uint32_t client_id = 0; bool ok = false; GameState state; uint64_t enqueue_ms = 0; };

    mutable std::mutex queue_mutex_;                       // pending_, finished_, next_emit_, callback_
    std::condition_variable queue_cv_;
    // Requests whose pixels are in the engine's ring, in the order zly_submit returned: the completion thread consumes them in
    // THIS order (never in sequence order: the owner of the next sequence number may be blocked in zly_submit by ring
    // back-pressure that only consuming later tickets releases), and re-orders the results by sequence number for the callback.
    // One queue per engine slot (+ one for requests without a ticket: refused or simulated): the completion thread drains whichever engine's batch is back
    // first.  One global queue made it consume tickets in push order, i.e. alternately from all engines: a finished batch of engine A stayed unconsumed --
    // its ring slot unrecycled, A's submitters back-pressured -- until the interleaved tickets of engine B's still running batch had been waited for.
    std::vector<std::deque<Pending>> pending_;
    size_t pending_count_ = 0;
    // results waiting for an earlier sequence number: a ring indexed by seq & (size - 1), grown when the span of outstanding sequence numbers exceeds it
    // (round 3: a std::map insert + erase per frame).  Touched by the completion thread under queue_mutex_ once per GROUP of completed frames.
    std::vector<Done> ring_;
    std::vector<uint8_t> ring_full_;
    size_t finished_count_ = 0;
    std::atomic<uint64_t> next_seq_{0};
    uint64_t next_emit_ = 0;
    InferenceCallback callback_;

    std::atomic<uint64_t> inference_count_{0}, inference_errors_{0}, dropped_frames_{0};
    std::atomic<size_t> queue_high_water_mark_{0};
    mutable std::mutex stats_mutex_;
    std::deque<double> latency_window_ms_;      // last 100 request latencies (onnx_engine.cpp:428-449)
    double total_latency_ms_ = 0;
};

class HipInferenceEngineFactory : public IInferenceEngineFactory {
public:
    std::unique_ptr<IInferenceEngine> createEngine(const ServerConfig& config) override;
    std::string getName() const override;
};

}  // namespace zero_latency
