// test_plugin_stub.cpp -- the plugin's HOST logic with several engines per process, without a GPU: round-robin over the engine handles,
// re-ordering of results that finish out of order, error frames that keep the sequence dense, hot reload with requests in flight, and WHERE
// replaced engines are destroyed.  TEST INFRASTRUCTURE: host/hip_inference_engine.cpp is compiled into this binary together with a
// link-time stub of the C-ABI entry points it calls (zly_create / zly_destroy / zly_submit / zly_submit_try / zly_poll / zly_wait / zly_get_stats / zly_weights_fp8 /
// zly_default_config / zly_last_error).  The stub is not a CPU fallback of the product: libzly.so has none, and this file is never linked
// into it.  A fake engine "detects" one box per frame whose fields encode (engine ordinal, device, first pixel of the frame), and
// finishes its tickets after a per-engine delay, so that engines complete out of order.
//
//   test_plugin_stub <report.txt>          (environment: ZLY_NUM_DEVICES=2 ZLY_ENGINES_PER_GPU=2 set by the driver itself)
#include "../../zero-latency-yolo_amd/host/hip_inference_engine.cpp"

#include <fstream>
#include <set>
#include <thread>

// ------------------------------------------------------------------------------------------------ stub of the C ABI
struct zly_engine {
    int ordinal = 0, device = 0, max_batch = 0;
    std::mutex mu;
    std::map<uint64_t, std::pair<uint8_t, std::chrono::steady_clock::time_point>> pending;     // ticket -> (first pixel, ready time)
    uint64_t next_ticket = 1;
    std::atomic<uint64_t> frames{0};
};
namespace {
std::mutex g_mu;
int g_created = 0, g_destroyed = 0, g_fail_create_after = -1;
std::vector<int> g_devices;                              // device ordinal of every zly_create
std::vector<std::thread::id> g_destroy_threads;          // thread of every zly_destroy
thread_local std::string g_err;
}

extern "C" {
void zly_default_config(zly_config* c) { std::memset(c, 0, sizeof *c); c->model_w = 416; c->model_h = 416; c->conf_thr = 0.5f; c->iou_thr = 0.45f; c->max_batch = 1; c->max_dets = 64; c->warmup_runs = 3; c->use_graph = 1; }
const char* zly_last_error(void) { return g_err.c_str(); }
int32_t zly_create(const zly_config* cfg, zly_engine** out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_fail_create_after >= 0 && g_created >= g_fail_create_after) { g_err = "stub: model load failed"; return ZLY_ERR_MODEL_LOAD; }
    zly_engine* e = new zly_engine();
    e->ordinal = g_created++; e->device = cfg->device; e->max_batch = cfg->max_batch;
    g_devices.push_back(cfg->device);
    *out = e;
    return ZLY_OK;
}
int32_t zly_destroy(zly_engine* e)
{
    std::this_thread::sleep_for(std::chrono::milliseconds(30));          // a real zly_destroy drains streams: it must not run on the completion thread
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_destroyed; g_destroy_threads.push_back(std::this_thread::get_id());
    delete e;
    return ZLY_OK;
}
int32_t zly_submit(zly_engine* e, const uint8_t* bgr, size_t nbytes, int32_t w, int32_t h, uint64_t* ticket)
{
    if (!bgr || nbytes != (size_t)w * h * 3) { g_err = "Invalid image data size"; return ZLY_ERR_INVALID_INPUT; }
    std::lock_guard<std::mutex> lk(e->mu);
    // engine k answers after (3 - k % 4) ms: later engines finish EARLIER than earlier ones
    const auto ready = std::chrono::steady_clock::now() + std::chrono::microseconds(300 * (3 - e->ordinal % 4) + 50);
    *ticket = e->next_ticket++;
    e->pending[*ticket] = std::make_pair(bgr[0], ready);
    return ZLY_OK;
}
int32_t zly_submit_try(zly_engine* e, const uint8_t* bgr, size_t nbytes, int32_t w, int32_t h, uint64_t* ticket) { return zly_submit(e, bgr, nbytes, w, h, ticket); }   // the stub's ring is never full
int32_t zly_poll(zly_engine* e, uint64_t ticket)
{
    std::lock_guard<std::mutex> lk(e->mu);
    auto it = e->pending.find(ticket);
    if (it == e->pending.end()) { g_err = "unknown ticket"; return ZLY_ERR_INVALID_ARGUMENT; }
    return std::chrono::steady_clock::now() >= it->second.second ? ZLY_OK : ZLY_PENDING;
}
int32_t zly_wait(zly_engine* e, uint64_t ticket, zly_det* out, int32_t cap, int32_t* n_out)
{
    std::pair<uint8_t, std::chrono::steady_clock::time_point> p;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        auto it = e->pending.find(ticket);
        if (it == e->pending.end()) { g_err = "unknown ticket"; return ZLY_ERR_INVALID_ARGUMENT; }
        p = it->second;
        e->pending.erase(it);
    }
    std::this_thread::sleep_until(p.second);
    if (cap >= 1) {
        std::memset(out, 0, sizeof *out);
        out[0].x = (float)e->ordinal; out[0].y = (float)e->device; out[0].w = (float)p.first; out[0].confidence = 0.75f; out[0].class_id = e->ordinal % 4;
    }
    *n_out = 1;
    e->frames++;
    return ZLY_OK;
}
int32_t zly_get_stats(const zly_engine* e, zly_stats* out) { std::memset(out, 0, sizeof *out); out->batches = e->frames.load(); out->sampled_frames = 0; return ZLY_OK; }
int32_t zly_weights_fp8(const zly_engine*) { return 0; }
}

// ------------------------------------------------------------------------------------------------ the test
using namespace zero_latency;

// ZLY_SIMULATE=1: the reference's simulation mode as an explicit opt-in (onnx_engine.cpp:1133-1177) -- and ONLY as that: a model that fails to load
// without the switch is an error from initialize(), never random boxes.
static int simulate_main(const char* report)
{
    std::ofstream rep(report);
    ServerConfig config;
    config.model_path = "/nonexistent/model.zlyw";
    config.inference_engine = "hip";
    // 1. no switch + a model that does not load: initialize() fails, nothing is created, nothing is simulated
    unsetenv("ZLY_SIMULATE");
    { std::lock_guard<std::mutex> lk(g_mu); g_fail_create_after = 0; }
    {
        auto engine = InferenceEngineManager::getInstance().createEngine("hip", config);
        auto r = engine->initialize();
        rep << "no_switch_init_error=" << (r.hasError() ? static_cast<int>(r.error().code) : 0) << "\nno_switch_status_sim=" << engine->getStatus()["simulation_mode"] << "\n";
        InferenceRequest q; q.width = 2; q.height = 2; q.data.assign(12, 1);
        rep << "no_switch_submit=" << static_cast<int>(engine->submitInference(q).error().code) << "\n";
    }
    // 2. the switch: no engine is created; every frame gets 0-5 random detections in the reference's ranges, in submission order
    setenv("ZLY_SIMULATE", "1", 1); setenv("ZLY_SIMULATE_SEED", "7", 1);
    std::vector<std::vector<float>> runs[2];
    for (int run = 0; run < 2; ++run) {
        auto engine = InferenceEngineManager::getInstance().createEngine("hip", config);
        std::mutex mu;
        std::vector<uint32_t> order;
        bool ranges = true;
        int maxn = 0, total = 0;
        engine->setCallback([&](uint32_t, const GameState& st) {
            std::lock_guard<std::mutex> lk(mu);
            order.push_back(st.frame_id);
            maxn = std::max(maxn, (int)st.detections.size());
            std::vector<float> flat;
            for (size_t i = 0; i < st.detections.size(); ++i) {
                const Detection& d = st.detections[i];
                ranges = ranges && d.box.x >= 0.1f && d.box.x <= 0.9f && d.box.y >= 0.1f && d.box.y <= 0.9f && d.box.width >= 0.05f && d.box.width <= 0.2f &&
                         d.box.height >= 0.075f && d.box.height <= 0.3f && d.confidence >= 0.6f && d.confidence <= 1.0f && d.class_id >= 0 && d.class_id <= 3 &&
                         d.track_id == (uint32_t)i + 1 && d.timestamp > 1600000000000ull;
                flat.push_back(d.box.x); flat.push_back(d.confidence); flat.push_back((float)d.class_id);
                ++total;
            }
            runs[run].push_back(flat);
        });
        if (engine->initialize().hasError()) return 4;
        for (int i = 0; i < 300; ++i) {
            InferenceRequest q; q.client_id = 1; q.frame_id = (uint32_t)i; q.timestamp = 5; q.width = 2; q.height = 2; q.data.assign(i % 7 == 0 ? 5 : 12, 1);   // the reference's mode never looks at the pixels
            if (engine->submitInference(q).hasError()) return 5;
        }
        for (int k = 0; k < 4000; ++k) { { std::lock_guard<std::mutex> lk(mu); if (order.size() >= 300) break; } std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
        auto st = engine->getStatus();
        engine->shutdown();
        std::lock_guard<std::mutex> lk(mu);
        bool in_order = order.size() == 300;
        for (size_t i = 0; i < order.size(); ++i) in_order = in_order && order[i] == (uint32_t)i;
        if (run == 0)
            rep << "sim_count=" << order.size() << "\nsim_in_order=" << in_order << "\nsim_ranges=" << ranges << "\nsim_max_per_frame=" << maxn << "\nsim_total=" << total
                << "\nsim_status=" << st["simulation_mode"] << "\nsim_status_count=" << st["inference_count"] << "\nsim_worker_threads=" << st["worker_threads"] << "\n";
    }
    rep << "sim_same_seed_same_boxes=" << (runs[0] == runs[1]) << "\n";
    { std::lock_guard<std::mutex> lk(g_mu); rep << "sim_engines_created=" << g_created << "\n"; }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    if (argc > 2 && std::string(argv[2]) == "simulate") return simulate_main(argv[1]);
    std::ofstream rep(argv[1]);
    setenv("ZLY_NUM_DEVICES", "2", 1); setenv("ZLY_ENGINES_PER_GPU", "2", 1); setenv("ZLY_MODEL_WATCH_MS", "0", 1);
    const std::string model = std::string(argv[1]) + ".model";
    { std::ofstream m(model); m << "weights v1"; }
    ServerConfig config;
    config.model_path = model;
    config.inference_engine = "hip";
    auto engine = InferenceEngineManager::getInstance().createEngine("hip", config);
    if (!engine) return 3;
    HipInferenceEngine* hip = dynamic_cast<HipInferenceEngine*>(engine.get());
    struct Seen { uint32_t client, frame; float ordinal, device, pixel; std::thread::id tid; };
    std::vector<Seen> seen;
    std::mutex smu;
    engine->setCallback([&](uint32_t client, const GameState& st) {
        std::lock_guard<std::mutex> lk(smu);
        const Detection& d = st.detections.at(0);
        seen.push_back(Seen{client, st.frame_id, d.box.x, d.box.y, d.box.width, std::this_thread::get_id()});
    });
    if (engine->initialize().hasError()) return 4;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        rep << "created=" << g_created << "\ndevices=";
        for (int d : g_devices) rep << d << ",";
        rep << "\n";
    }
    // ---- phase 1: one submitting thread, 64 frames (every 10th has a wrong byte count): global order, round robin, dense sequence -------
    const int N = 64;
    int accepted = 0;
    for (int i = 0; i < N; ++i) {
        InferenceRequest r;
        r.client_id = 9; r.frame_id = (uint32_t)i; r.timestamp = 1000 + (uint64_t)i; r.width = 4; r.height = 2;
        r.data.assign(4 * 2 * 3, (uint8_t)i);
        if (i % 10 == 7) r.data.pop_back();                               // INVALID_INPUT for this frame only: counted, no callback
        else ++accepted;
        if (engine->submitInference(r).hasError()) return 5;
    }
    auto wait_for = [&](size_t n) {
        for (int k = 0; k < 4000; ++k) {
            { std::lock_guard<std::mutex> lk(smu); if (seen.size() >= n) return true; }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        return false;
    };
    rep << "phase1_all=" << (wait_for((size_t)accepted) ? 1 : 0) << "\n";
    {
        std::lock_guard<std::mutex> lk(smu);
        bool in_order = true, rr = true, echo = true;
        std::set<std::thread::id> tids;
        uint32_t prev = 0; bool first = true;
        for (const Seen& s : seen) {
            if (!first && s.frame <= prev) in_order = false;
            prev = s.frame; first = false;
            // sequence number == frame id here (every submit takes one, failed or not): engine = seq % 4, device = engine / 2
            if ((int)s.ordinal != (int)(s.frame % 4) || (int)s.device != (int)(s.frame % 4) / 2) rr = false;
            if ((int)s.pixel != (int)s.frame || s.client != 9) echo = false;
            tids.insert(s.tid);
        }
        rep << "phase1_count=" << seen.size() << "\nphase1_in_order=" << in_order << "\nphase1_round_robin=" << rr << "\nphase1_echo=" << echo
            << "\nphase1_callback_threads=" << tids.size() << "\n";
    }
    auto st1 = engine->getStatus();
    rep << "status_errors=" << st1["inference_errors"] << "\nstatus_count=" << st1["inference_count"] << "\nstatus_devices=" << st1["devices"]
        << "\nstatus_engines_per_gpu=" << st1["engines_per_gpu"] << "\nstatus_worker_threads=" << st1["worker_threads"] << "\n";
    // ---- phase 2: four submitting threads + a hot reload in the middle -------------------------------------------------------------
    std::thread::id completion_tid;
    { std::lock_guard<std::mutex> lk(smu); completion_tid = seen.empty() ? std::thread::id() : seen[0].tid; seen.clear(); }
    std::atomic<int> submitted{0};
    std::vector<std::thread> subs;
    for (int t = 0; t < 4; ++t)
        subs.emplace_back([&, t] {
            for (int i = 0; i < 50; ++i) {
                InferenceRequest r;
                r.client_id = (uint32_t)t; r.frame_id = (uint32_t)i; r.timestamp = 1; r.width = 2; r.height = 2;
                r.data.assign(12, (uint8_t)(t * 50 + i));
                if (!engine->submitInference(r).hasError()) submitted++;
                if (t == 0 && i == 20) { { std::ofstream m(model); m << "weights v2"; } hip->reloadModel(); }
            }
        });
    for (auto& th : subs) th.join();
    rep << "phase2_all=" << (wait_for((size_t)submitted.load()) ? 1 : 0) << "\n";
    {
        std::lock_guard<std::mutex> lk(smu);
        bool per_client_order = true;
        std::map<uint32_t, int> last;
        int old_engines = 0, new_engines = 0;
        for (const Seen& s : seen) {
            if (last.count(s.client) && (int)s.frame <= last[s.client]) per_client_order = false;
            last[s.client] = (int)s.frame;
            if ((int)s.ordinal < 4) ++old_engines; else ++new_engines;
        }
        rep << "phase2_count=" << seen.size() << "\nphase2_per_client_order=" << per_client_order << "\nphase2_on_old_engines=" << (old_engines > 0)
            << "\nphase2_on_new_engines=" << (new_engines > 0) << "\n";
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(400));          // the reaper's periodic pass + 4 x 30 ms of stub teardown
    {
        std::lock_guard<std::mutex> lk(g_mu);
        bool on_completion = false;
        for (auto id : g_destroy_threads) if (id == completion_tid) on_completion = true;
        rep << "created_after_reload=" << g_created << "\ndestroyed_after_reload=" << g_destroyed << "\ndestroyed_on_completion_thread=" << on_completion << "\n";
    }
    rep << "model_version=" << engine->getStatus()["model_version"] << "\n";
    // ---- phase 3: a reload that fails leaves the running engines untouched ------------------------------------------------------------------
    { std::lock_guard<std::mutex> lk(g_mu); g_fail_create_after = g_created + 1; }     // the second of the four new engines fails
    auto bad = hip->reloadModel();
    { std::lock_guard<std::mutex> lk(g_mu); g_fail_create_after = -1; }
    rep << "failed_reload_code=" << static_cast<int>(bad.error().code) << "\nmodel_version_after_failed_reload=" << engine->getStatus()["model_version"] << "\n";
    { std::lock_guard<std::mutex> lk(smu); seen.clear(); }
    InferenceRequest r; r.client_id = 1; r.frame_id = 77; r.timestamp = 5; r.width = 2; r.height = 2; r.data.assign(12, 3);
    engine->submitInference(r);
    rep << "served_after_failed_reload=" << (wait_for(1) ? 1 : 0) << "\n";
    engine->shutdown();
    rep << "submit_after_shutdown=" << static_cast<int>(engine->submitInference(r).error().code) << "\n";
    {
        std::lock_guard<std::mutex> lk(g_mu);
        rep << "created_total=" << g_created << "\ndestroyed_total=" << g_destroyed << "\n";
    }
    return 0;
}
