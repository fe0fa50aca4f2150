// hip_inference_engine.cpp -- see hip_inference_engine.h.  Host C++ only; every GPU action is a call
// into the C ABI of include/zly.h.
#include "hip_inference_engine.h"

#include "zly.h"
#include "zly_sha256.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

namespace zero_latency {

namespace {
uint64_t wallMs()
{
    return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(
               std::chrono::system_clock::now().time_since_epoch()).count();
}
ErrorCode toErrorCode(int32_t rc)
{
    switch (rc) {
        case ZLY_OK: return ErrorCode::OK;
        case ZLY_ERR_INVALID_ARGUMENT: return ErrorCode::INVALID_ARGUMENT;
        case ZLY_ERR_NOT_INITIALIZED: return ErrorCode::NOT_INITIALIZED;
        case ZLY_ERR_MODEL_NOT_FOUND: return ErrorCode::MODEL_NOT_FOUND;
        case ZLY_ERR_MODEL_LOAD: return ErrorCode::MODEL_LOAD_FAILED;
        case ZLY_ERR_INVALID_INPUT: return ErrorCode::INVALID_INPUT;
        case ZLY_ERR_SYSTEM: return ErrorCode::SYSTEM_ERROR;
        default: return ErrorCode::INFERENCE_ERROR;
    }
}
int envInt(const char* name, int fallback)
{
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : fallback;
}
}  // namespace

struct HipInferenceEngine::EngineHandle {
    zly_engine* e = nullptr;
    explicit EngineHandle(zly_engine* p) : e(p) {}
    ~EngineHandle() { if (e) zly_destroy(e); }
    EngineHandle(const EngineHandle&) = delete;
    EngineHandle& operator=(const EngineHandle&) = delete;
};

HipInferenceEngine::HipInferenceEngine(const ServerConfig& config) : config_(config)
{
    max_batch_ = std::max(1, std::min(65535, envInt("ZLY_MAX_BATCH", 64)));
    max_dets_ = std::max(1, envInt("ZLY_MAX_DETS", 256));
}

HipInferenceEngine::~HipInferenceEngine() { shutdown(); }

Result<void> HipInferenceEngine::initialize()
{
    if (running_) return Result<void>::ok();
    const int ndev = std::max(1, envInt("ZLY_NUM_DEVICES", 1));
    const int dev0 = std::max(0, envInt("ZLY_FIRST_DEVICE", 0));
    first_device_ = dev0;
    // ZLY_ENGINES_PER_GPU (default 2): several engine instances per GPU, each one chain of launches on its own stream; consecutive
    // requests go to different instances and their batches overlap on the device (one instance leaves the chip idle at every kernel
    // boundary): 67k -> 82k frames/s host to host.  Two compute streams + the shared upload stream + the null stream = the four
    // hardware queues ROCm gives a process; a third instance shares a queue and is slower
    engines_per_gpu_ = std::max(1, std::min(8, envInt("ZLY_ENGINES_PER_GPU", 2)));
    std::vector<std::shared_ptr<EngineHandle>> fresh;
    for (int d = 0; d < ndev * engines_per_gpu_; ++d) {
        int32_t rc = ZLY_OK;
        std::string msg;
        auto e = createEngineOn(dev0 + d / engines_per_gpu_, &rc, &msg);
        if (!e) return Result<void>::error(toErrorCode(rc), "Failed to initialize HIP inference engine: " + msg);
        fresh.push_back(std::move(e));
    }
    {
        std::lock_guard<std::mutex> lk(engines_mutex_);
        engines_ = std::move(fresh);
    }
    {
        std::lock_guard<std::mutex> lk(stats_mutex_);
        model_hash_ = sha256File(config_.model_path);
    }
    model_version_ = 1;
    {
        std::lock_guard<std::mutex> lk(queue_mutex_);
        pending_.clear();                                      // nothing of a previous run may be emitted under the new sequence numbers
        finished_.clear();
        next_seq_ = 0;
        next_emit_ = 0;
    }
    running_ = true;
    {
        std::lock_guard<std::mutex> lk(reap_mutex_);
        reap_stop_ = false;
    }
    reaper_ = std::thread(&HipInferenceEngine::reaperLoop, this);
    completer_ = std::thread(&HipInferenceEngine::completionLoop, this);
    if (envInt("ZLY_MODEL_WATCH_MS", 10000) > 0) monitor_ = std::thread(&HipInferenceEngine::monitorLoop, this);
    return Result<void>::ok();
}

std::shared_ptr<HipInferenceEngine::EngineHandle> HipInferenceEngine::createEngineOn(int device, int32_t* rc, std::string* msg) const
{
    zly_config c;
    zly_default_config(&c);
    c.weights_path = config_.model_path.c_str();
    c.model_w = config_.detection.model_width;
    c.model_h = config_.detection.model_height;
    c.conf_thr = config_.confidence_threshold;
    c.iou_thr = config_.nms_threshold;
    c.max_batch = max_batch_;
    c.max_dets = max_dets_;
    c.device = device;
    c.dtype = envInt("ZLY_FP32", 0) ? ZLY_DTYPE_FP32 : ZLY_DTYPE_BF16;
    c.warmup_runs = 3;                                   // onnx_engine.cpp:919-954
    c.use_graph = 1;
    // the server only consumes detections; one engine per GPU: NMS of a batch runs beside the next one; several: one chain each
    c.flags = ZLY_FLAG_NO_HEAD_TENSOR | (engines_per_gpu_ > 1 ? ZLY_FLAG_SINGLE_CHAIN : ZLY_FLAG_ASYNC_NMS);
    zly_engine* e = nullptr;
    *rc = zly_create(&c, &e);
    if (*rc != ZLY_OK) { *msg = zly_last_error(); return nullptr; }
    return std::make_shared<EngineHandle>(e);
}

// Build the new engines first (hundreds of milliseconds: weight repack, upload, warm-up, beside the running ones), then
// swap the handles: requests submitted from then on go to the new model, requests already in flight finish on the old
// engine, which is destroyed with its last pending request.  Any failure leaves the running model untouched.
Result<void> HipInferenceEngine::reloadModel()
{
    if (!running_) return Result<void>::error(ErrorCode::NOT_INITIALIZED, "Engine not running");
    std::lock_guard<std::mutex> rl(reload_mutex_);
    const std::string hash = sha256File(config_.model_path);
    size_t count;
    {
        std::lock_guard<std::mutex> lk(engines_mutex_);
        count = engines_.size();
    }
    std::vector<std::shared_ptr<EngineHandle>> fresh;
    for (size_t d = 0; d < count; ++d) {
        int32_t rc = ZLY_OK;
        std::string msg;
        auto e = createEngineOn(first_device_ + (int)d / engines_per_gpu_, &rc, &msg);
        if (!e) return Result<void>::error(toErrorCode(rc), "Failed to reload model: " + msg);
        fresh.push_back(std::move(e));
    }
    {
        std::lock_guard<std::mutex> lk(engines_mutex_);
        engines_.swap(fresh);
    }
    for (auto& h : fresh) retire(std::move(h));                 // old engines go -- on the reaper thread -- when their last pending request is done
    fresh.clear();
    {
        std::lock_guard<std::mutex> lk(stats_mutex_);
        model_hash_ = hash;
    }
    model_version_++;
    return Result<void>::ok();
}

void HipInferenceEngine::retire(std::shared_ptr<EngineHandle>&& h)
{
    if (!h) return;
    const bool last = h.use_count() == 1;
    {
        std::lock_guard<std::mutex> lk(reap_mutex_);
        retired_.push_back(std::move(h));
    }
    if (last) reap_cv_.notify_one();                            // otherwise the reaper's next periodic pass drops it
}

void HipInferenceEngine::reaperLoop()
{
    std::vector<std::shared_ptr<EngineHandle>> batch;
    while (true) {
        {
            std::unique_lock<std::mutex> lk(reap_mutex_);
            reap_cv_.wait_for(lk, std::chrono::milliseconds(50), [&] { return reap_stop_ || !retired_.empty(); });
            batch.swap(retired_);
            if (batch.empty() && reap_stop_) return;
        }
        batch.clear();                                          // the last reference of a replaced engine: zly_destroy runs here
    }
}

void HipInferenceEngine::monitorLoop()
{
    const int period_ms = std::max(50, envInt("ZLY_MODEL_WATCH_MS", 10000));
    int waited = 0;
    while (running_) {
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        waited += 50;
        if (waited < period_ms) continue;
        waited = 0;
        const std::string now = sha256File(config_.model_path);
        if (now.empty()) continue;                              // file missing: keep serving (onnx_engine.cpp:484-488)
        std::string last;
        {
            std::lock_guard<std::mutex> lk(stats_mutex_);
            last = model_hash_;
        }
        if (now != last) {
            auto r = reloadModel();
            if (r.hasError()) {                                 // remember the bad file so it is not retried every period
                std::lock_guard<std::mutex> lk(stats_mutex_);
                model_hash_ = now;
            }
        }
    }
}

Result<void> HipInferenceEngine::shutdown()
{
    if (!running_.exchange(false)) {
        return Result<void>::ok();
    }
    {
        std::lock_guard<std::mutex> lk(queue_mutex_);
        queue_cv_.notify_all();
    }
    if (completer_.joinable()) completer_.join();               // hands over what is already on the device, then stops
    if (monitor_.joinable()) monitor_.join();
    {
        std::lock_guard<std::mutex> lk(reap_mutex_);
        reap_stop_ = true;
    }
    reap_cv_.notify_all();
    if (reaper_.joinable()) reaper_.join();
    {
        std::lock_guard<std::mutex> lk(queue_mutex_);
        dropped_frames_ += pending_.size();
        pending_.clear();                                       // stale results must never be emitted by a later initialize()
        finished_.clear();
    }
    std::vector<std::shared_ptr<EngineHandle>> old;
    {
        std::lock_guard<std::mutex> lk(engines_mutex_);
        old.swap(engines_);
    }
    old.clear();
    return Result<void>::ok();
}

// Runs on the caller's thread (the UDP receive thread in the reference's server, network_server.cpp:209): the request's
// pixels are copied ONCE, into the engine's pinned staging ring; nothing of `request` is retained.
Result<void> HipInferenceEngine::submitInference(const InferenceRequest& request)
{
    if (!running_) return Result<void>::error(ErrorCode::NOT_INITIALIZED, "Engine not running");
    const uint64_t seq = next_seq_.fetch_add(1);
    Pending p;
    p.seq = seq;
    p.client_id = request.client_id; p.frame_id = request.frame_id; p.timestamp = request.timestamp;
    p.enqueue_ms = wallMs();
    {
        std::lock_guard<std::mutex> lk(engines_mutex_);
        if (!engines_.empty()) p.engine = engines_[(size_t)(seq % engines_.size())];        // one-frame-per-engine round robin
    }
    if (!p.engine) {
        p.failed = true;
    } else {
        // a request with the wrong byte count fails alone (INVALID_INPUT, onnx_engine.cpp:659-665): counted, no callback
        const int32_t rc = zly_submit(p.engine->e, request.data.data(), request.data.size(), request.width, request.height, &p.ticket);
        if (rc != ZLY_OK) { p.failed = true; p.engine.reset(); }
    }
    if (p.failed) inference_errors_++;
    {
        std::lock_guard<std::mutex> lk(queue_mutex_);
        pending_.push_back(std::move(p));
        if (pending_.size() + finished_.size() > queue_high_water_mark_) queue_high_water_mark_ = pending_.size() + finished_.size();
    }
    queue_cv_.notify_one();
    return Result<void>::ok();
}

void HipInferenceEngine::setCallback(InferenceCallback callback)
{
    std::lock_guard<std::mutex> lk(queue_mutex_);
    callback_ = std::move(callback);
}

size_t HipInferenceEngine::getQueueSize() const
{
    std::lock_guard<std::mutex> lk(queue_mutex_);
    return pending_.size() + finished_.size();
}

std::string HipInferenceEngine::getName() const { return "hip"; }

// One engine-owned thread hands results to the callback in submission order (reference: the inference thread,
// onnx_engine.cpp:355-364); several GPUs finishing out of order are re-ordered by the sequence number.
void HipInferenceEngine::completionLoop()
{
    std::vector<zly_det> dets((size_t)max_dets_);
    while (true) {
        Pending p;
        {
            std::unique_lock<std::mutex> lk(queue_mutex_);
            queue_cv_.wait_for(lk, std::chrono::milliseconds(100), [&] { return !pending_.empty() || !running_; });
            if (pending_.empty()) {
                if (!running_) return;                           // shutting down: everything already in the ring has been handed over
                continue;
            }
            p = std::move(pending_.front());
            pending_.pop_front();
        }
        Done d;
        d.client_id = p.client_id; d.enqueue_ms = p.enqueue_ms;
        if (!p.failed) {
            int32_t n = 0;
            const int32_t rc = zly_wait(p.engine->e, p.ticket, dets.data(), max_dets_, &n);
            if (rc == ZLY_OK) {
                d.ok = true;
                d.state.frame_id = p.frame_id;                   // onnx_engine.cpp:520-521
                d.state.timestamp = p.timestamp;
                const int cnt = std::min<int>(n, max_dets_);
                d.state.detections.resize((size_t)cnt);
                static_assert(sizeof(zly_det) == sizeof(Detection), "zly_det must be layout-identical to Detection");
                if (cnt) std::memcpy(d.state.detections.data(), dets.data(), (size_t)cnt * sizeof(Detection));
            } else {
                inference_errors_++;
            }
            retire(std::move(p.engine));                         // an engine replaced by a reload goes with its last request -- on the reaper thread
        }
        // hand over in submission order
        std::vector<Done> ready;
        InferenceCallback cb;
        {
            std::lock_guard<std::mutex> lk(queue_mutex_);
            finished_.emplace(p.seq, std::move(d));
            while (!finished_.empty() && finished_.begin()->first == next_emit_) {
                ready.push_back(std::move(finished_.begin()->second));
                finished_.erase(finished_.begin());
                ++next_emit_;
            }
            cb = callback_;
        }
        for (Done& r : ready) {
            if (!r.ok) continue;                                 // not invoked on error results (onnx_engine.cpp:380-388)
            inference_count_++;
            {
                std::lock_guard<std::mutex> lk(stats_mutex_);
                const double lat = (double)(wallMs() - r.enqueue_ms);
                latency_window_ms_.push_back(lat);
                if (latency_window_ms_.size() > 100) latency_window_ms_.pop_front();
                total_latency_ms_ += lat;
            }
            if (cb) cb(r.client_id, r.state);
        }
    }
}

std::unordered_map<std::string, std::string> HipInferenceEngine::getStatus() const
{
    std::unordered_map<std::string, std::string> s;
    s["name"] = getName();
    s["simulation_mode"] = "false";
    s["running"] = running_ ? "true" : "false";
    s["model_path"] = config_.model_path;
    s["model_version"] = std::to_string(model_version_.load());
    { std::lock_guard<std::mutex> lk(stats_mutex_); s["model_hash"] = model_hash_; }
    s["int8_quantization"] = "disabled";
    s["zero_copy"] = "disabled";
    s["queue_size"] = std::to_string(getQueueSize());
    s["queue_high_water_mark"] = std::to_string(queue_high_water_mark_.load());
    s["inference_count"] = std::to_string(inference_count_.load());
    s["inference_errors"] = std::to_string(inference_errors_.load());
    s["dropped_frames"] = std::to_string(dropped_frames_.load());
    s["dynamic_batching"] = "enabled";
    double avg = 0, p99 = 0;
    {
        std::lock_guard<std::mutex> lk(stats_mutex_);
        if (!latency_window_ms_.empty()) {
            std::vector<double> v(latency_window_ms_.begin(), latency_window_ms_.end());
            std::sort(v.begin(), v.end());
            for (double x : v) avg += x;
            avg /= (double)v.size();
            p99 = v[std::min(v.size() - 1, (size_t)(v.size() * 0.99))];
        }
    }
    s["avg_inference_time_ms"] = std::to_string(avg);
    s["p99_inference_time_ms"] = std::to_string(p99);
    // per-phase device time, sampled by the engine on every 16th batch (the reference accumulates its three phase
    // timers per frame, onnx_engine.cpp:530-557,605-618).  zly_get_stats never waits on a running batch.
    std::vector<std::shared_ptr<EngineHandle>> engines;
    {
        std::lock_guard<std::mutex> lk(engines_mutex_);
        engines = engines_;
    }
    uint64_t frames = 0, batches = 0, replays = 0, eager = 0;
    double pre = 0, fwd = 0, post = 0;
    for (const auto& h : engines) {
        zly_stats st{};
        if (zly_get_stats(h->e, &st) != ZLY_OK) continue;
        frames += st.sampled_frames; batches += st.batches; replays += st.graph_replays; eager += st.eager_batches;
        pre += st.sampled_preprocess_ms; fwd += st.sampled_forward_ms; post += st.sampled_postprocess_ms;
    }
    s["batches"] = std::to_string(batches);
    s["graph_replays"] = std::to_string(replays);          // batches whose forward was one hipGraph replay (lone frames and full batches) ...
    s["eager_batches"] = std::to_string(eager);            // ... or kernel-by-kernel launches (partial batches)
    s["avg_preprocessing_time_ms"] = frames ? std::to_string(pre / (double)frames) : "0";
    s["avg_forward_time_ms"] = frames ? std::to_string(fwd / (double)frames) : "0";
    s["avg_postprocessing_time_ms"] = frames ? std::to_string(post / (double)frames) : "0";
    s["weight_format"] = (!engines.empty() && zly_weights_fp8(engines[0]->e)) ? "fp8_e4m3" : "fp32";      // the reference's "quantised" model claim (README.md:84)
    s["worker_threads"] = std::to_string(engines.size());
    s["devices"] = std::to_string(engines.size() / (size_t)std::max(1, engines_per_gpu_));
    s["engines_per_gpu"] = std::to_string(engines_per_gpu_);
    return s;
}

std::unique_ptr<IInferenceEngine> HipInferenceEngineFactory::createEngine(const ServerConfig& config)
{
    return std::make_unique<HipInferenceEngine>(config);
}

std::string HipInferenceEngineFactory::getName() const { return "hip"; }

REGISTER_INFERENCE_ENGINE(HipInferenceEngineFactory)

}  // namespace zero_latency
