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

HipInferenceEngine::HipInferenceEngine(const ServerConfig& config) : config_(config)
{
    max_batch_ = std::max(1, envInt("ZLY_MAX_BATCH", 8));
    max_dets_ = std::max(1, envInt("ZLY_MAX_DETS", 256));
}

HipInferenceEngine::~HipInferenceEngine() { shutdown(); }

Result<void> HipInferenceEngine::initialize()
{
    if (running_) return Result<void>::ok();
    const int ndev = std::max(1, envInt("ZLY_NUM_DEVICES", 1));
    const int dev0 = std::max(0, envInt("ZLY_FIRST_DEVICE", 0));
    first_device_ = dev0;
    for (int d = 0; d < ndev; ++d) {
        int32_t rc = ZLY_OK;
        std::string msg;
        zly_engine* e = createEngineOn(dev0 + d, &rc, &msg);
        if (!e) {
            for (zly_engine* p : engines_) zly_destroy(p);
            engines_.clear();
            engine_mutex_.clear();
            return Result<void>::error(toErrorCode(rc), "Failed to initialize HIP inference engine: " + msg);
        }
        engines_.push_back(e);
        engine_mutex_.emplace_back(new std::mutex);
    }
    {
        std::lock_guard<std::mutex> lk(stats_mutex_);
        model_hash_ = sha256File(config_.model_path);
    }
    model_version_ = 1;
    running_ = true;
    next_seq_ = 0;
    next_emit_ = 0;
    for (int d = 0; d < ndev; ++d) workers_.emplace_back(&HipInferenceEngine::workerLoop, this, d);
    if (envInt("ZLY_MODEL_WATCH_MS", 10000) > 0) monitor_ = std::thread(&HipInferenceEngine::monitorLoop, this);
    return Result<void>::ok();
}

zly_engine* HipInferenceEngine::createEngineOn(int device, int32_t* rc, std::string* msg) const
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
    c.flags = ZLY_FLAG_NO_HEAD_TENSOR;                   // the server only consumes detections
    zly_engine* e = nullptr;
    *rc = zly_create(&c, &e);
    if (*rc != ZLY_OK) { *msg = zly_last_error(); return nullptr; }
    return e;
}

// Build the new engines first (hundreds of milliseconds: weight repack, upload, warm-up, beside the running ones), then
// swap each one in while its worker is between two batches.  Any failure leaves the running model untouched.
Result<void> HipInferenceEngine::reloadModel()
{
    if (!running_) return Result<void>::error(ErrorCode::NOT_INITIALIZED, "Engine not running");
    std::lock_guard<std::mutex> rl(reload_mutex_);
    const std::string hash = sha256File(config_.model_path);
    std::vector<zly_engine*> fresh;
    for (size_t d = 0; d < engines_.size(); ++d) {
        int32_t rc = ZLY_OK;
        std::string msg;
        zly_engine* e = createEngineOn(first_device_ + (int)d, &rc, &msg);
        if (!e) {
            for (zly_engine* p : fresh) zly_destroy(p);
            return Result<void>::error(toErrorCode(rc), "Failed to reload model: " + msg);
        }
        fresh.push_back(e);
    }
    for (size_t d = 0; d < engines_.size(); ++d) {
        zly_engine* old = nullptr;
        {
            std::lock_guard<std::mutex> el(*engine_mutex_[d]);
            old = engines_[d];
            engines_[d] = fresh[d];
        }
        zly_destroy(old);
    }
    {
        std::lock_guard<std::mutex> lk(stats_mutex_);
        model_hash_ = hash;
    }
    model_version_++;
    return Result<void>::ok();
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
    for (std::thread& t : workers_)
        if (t.joinable()) t.join();
    workers_.clear();
    if (monitor_.joinable()) monitor_.join();
    for (zly_engine* e : engines_) zly_destroy(e);
    engines_.clear();
    engine_mutex_.clear();
    {
        std::lock_guard<std::mutex> lk(queue_mutex_);
        dropped_frames_ += queue_.size();
        queue_.clear();
    }
    return Result<void>::ok();
}

Result<void> HipInferenceEngine::submitInference(const InferenceRequest& request)
{
    if (!running_) return Result<void>::error(ErrorCode::NOT_INITIALIZED, "Engine not running");
    {
        std::lock_guard<std::mutex> lk(queue_mutex_);
        queue_.push_back(Pending{next_seq_++, request, wallMs()});       // copies the pixels: the caller's request is not retained
        if (queue_.size() > queue_high_water_mark_) queue_high_water_mark_ = queue_.size();
    }
    queue_cv_.notify_one();
    return Result<void>::ok();
}

void HipInferenceEngine::setCallback(InferenceCallback callback)
{
    std::lock_guard<std::mutex> lk(emit_mutex_);
    callback_ = std::move(callback);
}

size_t HipInferenceEngine::getQueueSize() const
{
    std::lock_guard<std::mutex> lk(queue_mutex_);
    return queue_.size();
}

std::string HipInferenceEngine::getName() const { return "hip"; }

void HipInferenceEngine::workerLoop(int worker)
{
    std::vector<Pending> batch;
    std::vector<const uint8_t*> ptrs;
    std::vector<size_t> nbytes;
    std::vector<int32_t> ws, hs, n_out;
    std::vector<zly_det> dets((size_t)max_batch_ * max_dets_);
    while (true) {
        batch.clear();
        {
            std::unique_lock<std::mutex> lk(queue_mutex_);
            queue_cv_.wait(lk, [&] { return !running_ || !queue_.empty(); });
            if (!running_) return;
            // take whatever is pending, up to one batch: no batching window, so a lone request is
            // served at single-frame latency and a backlog is served at batch throughput
            while (!queue_.empty() && (int)batch.size() < max_batch_) {
                batch.push_back(std::move(queue_.front()));
                queue_.pop_front();
            }
        }
        const int n = (int)batch.size();
        std::vector<std::pair<uint64_t, Done>> finished;
        finished.reserve((size_t)n);
        // a request with the wrong byte count fails alone (INVALID_INPUT, onnx_engine.cpp:659-665)
        std::vector<int> good;
        for (int i = 0; i < n; ++i) {
            const InferenceRequest& r = batch[(size_t)i].request;
            if (r.width == 0 || r.height == 0 || r.data.size() != (size_t)r.width * r.height * 3u) {
                inference_errors_++;
                finished.emplace_back(batch[(size_t)i].seq, Done{r.client_id, false, GameState{}});
            } else {
                good.push_back(i);
            }
        }
        if (!good.empty()) {
            const int m = (int)good.size();
            ptrs.resize((size_t)m); nbytes.resize((size_t)m); ws.resize((size_t)m); hs.resize((size_t)m); n_out.assign((size_t)m, 0);
            for (int k = 0; k < m; ++k) {
                const InferenceRequest& r = batch[(size_t)good[(size_t)k]].request;
                ptrs[(size_t)k] = r.data.data(); nbytes[(size_t)k] = r.data.size(); ws[(size_t)k] = r.width; hs[(size_t)k] = r.height;
            }
            int32_t rc;
            {
                std::lock_guard<std::mutex> el(*engine_mutex_[(size_t)worker]);      // a reload swaps the handle between two batches
                rc = zly_detect_batch(engines_[(size_t)worker], m, ptrs.data(), nbytes.data(), ws.data(), hs.data(), dets.data(), max_dets_, n_out.data());
            }
            batches_++;
            const uint64_t done_ms = wallMs();
            for (int k = 0; k < m; ++k) {
                const Pending& p = batch[(size_t)good[(size_t)k]];
                Done d;
                d.client_id = p.request.client_id;
                d.ok = rc == ZLY_OK;
                if (d.ok) {
                    d.state.frame_id = p.request.frame_id;                 // onnx_engine.cpp:520-521
                    d.state.timestamp = p.request.timestamp;
                    const int cnt = std::min<int>(n_out[(size_t)k], max_dets_);
                    d.state.detections.resize((size_t)cnt);
                    static_assert(sizeof(zly_det) == sizeof(Detection), "zly_det must be layout-identical to Detection");
                    if (cnt) std::memcpy(d.state.detections.data(), dets.data() + (size_t)k * max_dets_, (size_t)cnt * sizeof(Detection));
                    inference_count_++;
                    std::lock_guard<std::mutex> lk(stats_mutex_);
                    const double lat = (double)(done_ms - p.enqueue_ms);
                    latency_window_ms_.push_back(lat);
                    if (latency_window_ms_.size() > 100) latency_window_ms_.pop_front();
                    total_latency_ms_ += lat;
                } else {
                    inference_errors_++;
                }
                finished.emplace_back(p.seq, std::move(d));
            }
        }
        emitInOrder(std::move(finished));
    }
}

// Callbacks fire in submission order even when several GPUs finish out of order.
void HipInferenceEngine::emitInOrder(std::vector<std::pair<uint64_t, Done>>&& finished)
{
    std::lock_guard<std::mutex> lk(emit_mutex_);
    for (auto& f : finished) finished_.emplace(f.first, std::move(f.second));
    while (!finished_.empty() && finished_.begin()->first == next_emit_) {
        Done& d = finished_.begin()->second;
        if (d.ok && callback_) callback_(d.client_id, d.state);             // not invoked on error results (:380-388)
        finished_.erase(finished_.begin());
        ++next_emit_;
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
    s["batches"] = std::to_string(batches_.load());
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
    zly_stats st{};
    bool have = false;
    if (!engines_.empty() && !engine_mutex_.empty()) {
        std::lock_guard<std::mutex> el(*engine_mutex_[0]);          // the handle may be swapped by a reload
        have = zly_get_stats(engines_[0], &st) == ZLY_OK;
    }
    if (have && st.inference_count > 0) {
        s["avg_preprocessing_time_ms"] = std::to_string(st.total_preprocess_ms / (double)st.inference_count);
        s["avg_postprocessing_time_ms"] = std::to_string(st.total_postprocess_ms / (double)st.inference_count);
    } else {
        s["avg_preprocessing_time_ms"] = "0";
        s["avg_postprocessing_time_ms"] = "0";
    }
    s["worker_threads"] = std::to_string(workers_.size());
    s["devices"] = std::to_string(engines_.size());
    return s;
}

std::unique_ptr<IInferenceEngine> HipInferenceEngineFactory::createEngine(const ServerConfig& config)
{
    return std::make_unique<HipInferenceEngine>(config);
}

std::string HipInferenceEngineFactory::getName() const { return "hip"; }

REGISTER_INFERENCE_ENGINE(HipInferenceEngineFactory)

}  // namespace zero_latency
