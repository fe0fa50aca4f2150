// hip_inference_engine.cpp -- see hip_inference_engine.h.  Host C++ only; every GPU action is a call
// into the C ABI of include/zly.h.
#include "hip_inference_engine.h"

#include "zly.h"
#include "zly_sha256.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <random>

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
    std::atomic<bool> replaced{false};                   // set by reloadModel / shutdown when the handle leaves engines_: its last request hands it to the reaper
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
    simulate_ = envInt("ZLY_SIMULATE", 0) != 0;               // explicit opt-in only: a missing or bad model file is an ERROR below, never a silent fake
    const int ndev = simulate_ ? 0 : std::max(1, envInt("ZLY_NUM_DEVICES", 1));
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
        std::atomic_store(&engines_snapshot_, std::make_shared<const std::vector<std::shared_ptr<EngineHandle>>>(engines_));
    }
    {
        std::lock_guard<std::mutex> lk(stats_mutex_);
        model_hash_ = simulate_ ? std::string() : sha256File(config_.model_path);
    }
    if (simulate_) {
        // std::random_device like the reference (:1137-1138), or a fixed seed for tests (ZLY_SIMULATE_SEED)
        std::random_device rd;
        const int seed = envInt("ZLY_SIMULATE_SEED", -1);
        for (int i = 0; i < 4; ++i) sim_rng_state_[i] = seed >= 0 ? (uint32_t)seed * 2654435761u + (uint32_t)i : rd();
    }
    model_version_ = 1;
    {
        std::lock_guard<std::mutex> lk(queue_mutex_);
        pending_.assign((size_t)(ndev * engines_per_gpu_) + 1, std::deque<Pending>());     // nothing of a previous run may be emitted under the new sequence numbers
        pending_count_ = 0;
        ring_.assign(1024, Done{});
        ring_full_.assign(1024, 0);
        finished_count_ = 0;
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
    if (!simulate_ && envInt("ZLY_MODEL_WATCH_MS", 10000) > 0) monitor_ = std::thread(&HipInferenceEngine::monitorLoop, this);
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
    if (simulate_) return Result<void>::error(ErrorCode::INVALID_ARGUMENT, "simulation mode has no model to reload");
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
        std::atomic_store(&engines_snapshot_, std::make_shared<const std::vector<std::shared_ptr<EngineHandle>>>(engines_));
    }
    for (auto& h : fresh) { h->replaced = true; retire(std::move(h)); }     // old engines go -- on the reaper thread -- when their last pending request is done
    fresh.clear();
    {
        std::lock_guard<std::mutex> lk(stats_mutex_);
        model_hash_ = hash;
    }
    model_version_++;
    return Result<void>::ok();
}

// Engines replaced by a hot reload: the reaper takes a reference of each and KEEPS it until it is the only one left (requests in flight on the old
// engine hold the others), then runs zly_destroy -- on its own thread, never on the completion thread or a submitting thread.
void HipInferenceEngine::retire(std::shared_ptr<EngineHandle>&& h)
{
    if (!h) return;
    {
        std::lock_guard<std::mutex> lk(reap_mutex_);
        retired_.push_back(std::move(h));
    }
    reap_cv_.notify_one();
}

// A finished request lets go of its engine.  That is an atomic decrement and never the last reference: engines_ holds one while the engine serves, the
// reaper holds one from the moment a reload replaces it until every request has let go.  (Round 3 pushed EVERY frame's reference through reap_mutex_
// and the reaper's vector.)
void HipInferenceEngine::release(std::shared_ptr<EngineHandle>&& h) { h.reset(); }

void HipInferenceEngine::reaperLoop()
{
    std::vector<std::shared_ptr<EngineHandle>> mine, dead;
    while (true) {
        bool stop;
        {
            std::unique_lock<std::mutex> lk(reap_mutex_);
            reap_cv_.wait_for(lk, std::chrono::milliseconds(mine.empty() ? 50 : 5), [&] { return reap_stop_ || !retired_.empty(); });
            for (auto& h : retired_) mine.push_back(std::move(h));
            retired_.clear();
            stop = reap_stop_;
        }
        // sole owner = the last request on that engine has been delivered: destroy it here.  At shutdown the completion thread has been joined already;
        // what is still referenced elsewhere (requests dropped by shutdown) is let go and dies with its last reference on the shutting-down thread.
        for (auto& h : mine)
            if (stop || h.use_count() == 1) dead.push_back(std::move(h));
        mine.erase(std::remove(mine.begin(), mine.end(), nullptr), mine.end());
        dead.clear();                                           // zly_destroy runs here
        if (stop) return;
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
        dropped_frames_ += pending_count_;
        for (auto& q : pending_) q.clear();                     // stale results must never be emitted by a later initialize()
        pending_count_ = 0;
        ring_.clear(); ring_full_.clear(); finished_count_ = 0;
    }
    std::vector<std::shared_ptr<EngineHandle>> old;
    {
        std::lock_guard<std::mutex> lk(engines_mutex_);
        old.swap(engines_);
        std::atomic_store(&engines_snapshot_, std::shared_ptr<const std::vector<std::shared_ptr<EngineHandle>>>());
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
    if (simulate_) {
        // the reference's simulation mode checks nothing about the frame (runInference, :524-528)
        p.simulated = true;
    } else {
        const auto snap = std::atomic_load(&engines_snapshot_);                           // immutable snapshot: no lock shared by the submitting threads
        if (!snap || snap->empty()) {
            p.failed = true;
        } else {
            // One frame per engine, round robin -- offered without blocking first: when the engine whose turn it is has every ring slot
            // busy, the frame goes to the next one that has room, and only if all are back-pressured does the caller wait (on the engine whose turn
            // it was).  With the blocking call alone all submitting threads ended up waiting on ONE engine's ring while the other idled (round 4:
            // plugin 72k -> 77k frames/s against the C ABI's 86-94k with threads pinned to engines).  Results are re-ordered by sequence number.
            // A request with the wrong byte count fails alone (INVALID_INPUT, onnx_engine.cpp:659-665): counted, no callback.
            const size_t ne = snap->size(), first = (size_t)(seq % ne);
            int32_t rc = ZLY_PENDING;
            for (size_t k = 0; k < ne && rc == ZLY_PENDING; ++k) {
                p.slot = (first + k) % ne;
                p.engine = (*snap)[p.slot];
                rc = zly_submit_try(p.engine->e, request.data.data(), request.data.size(), request.width, request.height, &p.ticket);
            }
            if (rc == ZLY_PENDING) {
                p.slot = first;
                p.engine = (*snap)[first];
                rc = zly_submit(p.engine->e, request.data.data(), request.data.size(), request.width, request.height, &p.ticket);
            }
            if (rc != ZLY_OK) { p.failed = true; p.engine.reset(); }
        }
        if (p.failed) inference_errors_++;
    }
    bool wake;
    {
        std::lock_guard<std::mutex> lk(queue_mutex_);
        if (pending_.empty()) return Result<void>::error(ErrorCode::NOT_INITIALIZED, "Engine not running");     // shutdown() got in between
        wake = pending_count_ == 0;                              // the completion thread only sleeps on empty queues
        const size_t qi = (p.failed || p.simulated) ? pending_.size() - 1 : std::min(p.slot, pending_.size() - 2);
        pending_[qi].push_back(std::move(p));
        ++pending_count_;
        const size_t depth = (size_t)(seq + 1 - next_emit_);
        if (depth > queue_high_water_mark_) queue_high_water_mark_ = depth;
    }
    if (wake) queue_cv_.notify_one();
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
    return (size_t)(next_seq_.load() - next_emit_);              // submitted, not yet handed to the callback (or dropped as failed)
}

std::string HipInferenceEngine::getName() const { return "hip"; }

// generateRandomDetections of the reference's simulation mode (onnx_engine.cpp:1133-1177), ZLY_SIMULATE=1 only: 0-5 detections, centre in [0.1, 0.9],
// width in [0.05, 0.2], height = 1.5 x a second draw from that range, confidence in [0.6, 1.0], class in 0..3, track_id = i + 1, timestamp = now (ms).
std::vector<Detection> HipInferenceEngine::generateRandomDetections()
{
    auto next = [this]() {                                       // xoshiro128**: small, seedable state owned by the completion thread
        uint32_t* st = sim_rng_state_;
        const uint32_t x = st[1] * 5u, r = ((x << 7) | (x >> 25)) * 9u, t = st[1] << 9;
        st[2] ^= st[0]; st[3] ^= st[1]; st[1] ^= st[2]; st[0] ^= st[3]; st[2] ^= t; st[3] = (st[3] << 11) | (st[3] >> 21);
        return r;
    };
    auto uni = [&](double lo, double hi) { return lo + (hi - lo) * ((double)(next() >> 8) * (1.0 / 16777216.0)); };
    std::vector<Detection> out((size_t)(next() % 6u));
    const uint64_t now = wallMs();
    for (size_t i = 0; i < out.size(); ++i) {
        Detection& d = out[i];
        d.box.x = (float)uni(0.1, 0.9); d.box.y = (float)uni(0.1, 0.9);
        d.box.width = (float)uni(0.05, 0.2); d.box.height = (float)uni(0.05, 0.2) * 1.5f;
        d.confidence = (float)uni(0.6, 1.0);
        d.class_id = (int)(next() % 4u);
        d.track_id = (uint32_t)i + 1;
        d.timestamp = now;
    }
    return out;
}

// One engine-owned thread hands results to the callback in submission order (reference: the inference thread,
// onnx_engine.cpp:355-364); several GPUs finishing out of order are re-ordered by the sequence number.
// Per wake-up it takes EVERYTHING that is pending (one lock), blocks for the first ticket and then collects, without blocking, every following
// ticket whose batch is back as well (zly_poll): a device batch of 64 frames costs two lock acquisitions on queue_mutex_ and one on stats_mutex_,
// where round 3 paid two + one per FRAME against twelve submitting threads on the same mutex.
void HipInferenceEngine::completionLoop()
{
    std::vector<zly_det> dets((size_t)max_dets_);
    std::vector<std::deque<Pending>> local;                     // per engine slot (the last queue: requests without a ticket), in ticket order
    size_t local_count = 0;
    std::vector<std::pair<uint64_t, Done>> group;               // (sequence number, result) of the frames completed in this round
    std::vector<Done> ready;
    auto take = [&](Pending& p) {                               // consume one request whose result is available (or whose batch this call waits for)
        Done d;
        d.client_id = p.client_id; d.enqueue_ms = p.enqueue_ms;
        if (p.simulated) {
            d.ok = true;
            d.state.frame_id = p.frame_id; d.state.timestamp = p.timestamp;
            d.state.detections = generateRandomDetections();
        } else if (!p.failed) {
            int32_t n = 0;
            const int32_t rc = zly_wait(p.engine->e, p.ticket, dets.data(), max_dets_, &n);
            if (rc == ZLY_OK) {
                d.ok = true;
                d.state.frame_id = p.frame_id;                   // onnx_engine.cpp:520-521
                d.state.timestamp = p.timestamp;
                const int cnt = std::min<int>(n, max_dets_);
                static_assert(sizeof(zly_det) == sizeof(Detection), "zly_det must be layout-identical to Detection");
                d.state.detections.resize((size_t)cnt);
                if (cnt) std::memcpy(d.state.detections.data(), dets.data(), (size_t)cnt * sizeof(Detection));
            } else {
                inference_errors_++;
            }
            release(std::move(p.engine));                        // an engine replaced by a reload goes with its last request -- on the reaper thread
        }
        group.emplace_back(p.seq, std::move(d));
    };
    while (true) {
        {
            std::unique_lock<std::mutex> lk(queue_mutex_);
            if (local_count == 0) {
                queue_cv_.wait_for(lk, std::chrono::milliseconds(100), [&] { return pending_count_ != 0 || !running_; });
                if (pending_count_ == 0) {
                    if (!running_) return;                       // shutting down: everything already in the rings has been handed over
                    continue;
                }
            }
            if (local.size() != pending_.size()) local.resize(pending_.size());
            for (size_t i = 0; i < pending_.size(); ++i)         // everything that is pending, under ONE lock
                while (!pending_[i].empty()) { local[i].push_back(std::move(pending_[i].front())); pending_[i].pop_front(); ++local_count; }
            pending_count_ = 0;
        }
        group.clear();
        // every engine's tickets whose batch is back, without blocking (requests without a ticket are always ready)
        for (size_t i = 0; i < local.size(); ++i)
            while (!local[i].empty()) {
                Pending& p = local[i].front();
                if (!p.failed && !p.simulated && zly_poll(p.engine->e, p.ticket) != ZLY_OK) break;
                take(p);
                local[i].pop_front(); --local_count;
            }
        if (group.empty() && local_count != 0) {
            // nothing is back yet: wait for the oldest request's batch (the one the callback order needs first)
            size_t best = local.size();
            for (size_t i = 0; i < local.size(); ++i)
                if (!local[i].empty() && (best == local.size() || local[i].front().seq < local[best].front().seq)) best = i;
            take(local[best].front());
            local[best].pop_front(); --local_count;
        }
        // hand over in submission order
        ready.clear();
        InferenceCallback cb;
        {
            std::lock_guard<std::mutex> lk(queue_mutex_);
            for (auto& g : group) {
                while (g.first - next_emit_ >= ring_.size()) {   // more sequence numbers outstanding than the ring holds: double it (re-placing what it holds)
                    std::vector<Done> bigger(ring_.size() * 2);
                    std::vector<uint8_t> full(ring_.size() * 2, 0);
                    for (uint64_t q = next_emit_; q < next_emit_ + ring_.size(); ++q)
                        if (ring_full_[q & (ring_.size() - 1)]) { bigger[q & (bigger.size() - 1)] = std::move(ring_[q & (ring_.size() - 1)]); full[q & (bigger.size() - 1)] = 1; }
                    ring_.swap(bigger); ring_full_.swap(full);
                }
                const size_t at = (size_t)(g.first & (ring_.size() - 1));
                ring_[at] = std::move(g.second); ring_full_[at] = 1; ++finished_count_;
            }
            while (ring_full_[next_emit_ & (ring_.size() - 1)]) {
                const size_t at = (size_t)(next_emit_ & (ring_.size() - 1));
                ready.push_back(std::move(ring_[at]));
                ring_full_[at] = 0; --finished_count_;
                ++next_emit_;
            }
            cb = callback_;
        }
        if (ready.empty()) continue;
        {
            std::lock_guard<std::mutex> lk(stats_mutex_);
            const uint64_t now = wallMs();
            for (const Done& r : ready) {
                if (!r.ok) continue;
                const double lat = (double)(now - r.enqueue_ms);
                latency_window_ms_.push_back(lat);
                if (latency_window_ms_.size() > 100) latency_window_ms_.pop_front();
                total_latency_ms_ += lat;
            }
        }
        for (Done& r : ready) {
            if (!r.ok) continue;                                 // not invoked on error results (onnx_engine.cpp:380-388)
            inference_count_++;
            if (cb) cb(r.client_id, r.state);
        }
    }
}

std::unordered_map<std::string, std::string> HipInferenceEngine::getStatus() const
{
    std::unordered_map<std::string, std::string> s;
    s["name"] = getName();
    s["simulation_mode"] = simulate_ ? "true" : "false";
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
