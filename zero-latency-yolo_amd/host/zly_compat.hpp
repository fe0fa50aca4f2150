// zly_compat.hpp -- the declarations a plugin needs from the reference's headers, in one
// self-contained, compilable header.
//
// Why this exists: the plugin boundary of the reference is src/inference/inference_engine.h:16-103,
// but that header cannot be included here or anywhere -- it pulls in server/config.h (un-vendored
// nlohmann/json), common/memory_pool.h and common/event_bus.h (which do not compile), and the tree
// defines ErrorCode twice.  This header re-declares, from scratch and with the same
// names, namespaces, member names and signatures, exactly the types that cross the boundary:
//
//   BoundingBox, Detection, GameState            reference src/common/types.h:16-40 (Detection is
//                                                memcpy'd onto the wire, protocol.h:563-566: layout is ABI)
//   ErrorCode, Error, Result<T>                  reference src/common/result.h:14-221 (subset of the API
//                                                that callers of the engine use)
//   ServerConfig / DetectionConfig               reference src/server/config.h:110-149,305-345 (only the
//                                                keys the engine consumes)
//   InferenceRequest, InferenceCallback          reference src/inference/inference_engine.h:16-31
//   IInferenceEngine, IInferenceEngineFactory    :33-50
//   InferenceEngineManager, REGISTER_INFERENCE_ENGINE   :52-103
//
// In the reference tree a maintainer deletes this file and includes the real headers instead
// (INTEGRATION.md); nothing else in host/ changes.
#pragma once

#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

namespace zero_latency {

// ---- common/types.h ------------------------------------------------------------------------------
struct BoundingBox {
    float x, y, width, height;     // centre-x, centre-y, w, h, normalised by the request's dims
};

struct Detection {
    BoundingBox box;
    float confidence;
    int class_id;
    uint32_t track_id;
    uint64_t timestamp;
};
static_assert(sizeof(Detection) == 40, "Detection is copied raw into DetectionResultPacket");

struct GameState {
    uint32_t frame_id;
    uint64_t timestamp;
    std::vector<Detection> detections;
};

// ---- common/result.h -----------------------------------------------------------------------------
enum class ErrorCode {
    OK = 0, UNKNOWN_ERROR = 1, INVALID_ARGUMENT = 2, NOT_INITIALIZED = 3, TIMEOUT = 4,
    NETWORK_ERROR = 100, INVALID_PACKET = 103, PACKET_TOO_LARGE = 104, PROTOCOL_ERROR = 105,
    INFERENCE_ERROR = 200, MODEL_NOT_FOUND = 201, MODEL_LOAD_FAILED = 202, INVALID_INPUT = 203, INFERENCE_TIMEOUT = 204,
    SYSTEM_ERROR = 300, FILE_NOT_FOUND = 301, INSUFFICIENT_RESOURCES = 303,
    CONFIG_ERROR = 400
};

struct Error {
    ErrorCode code = ErrorCode::OK;
    std::string message;
    Error() = default;
    Error(ErrorCode c, std::string m) : code(c), message(std::move(m)) {}
    bool isOk() const { return code == ErrorCode::OK; }
    std::string toString() const { return "Error " + std::to_string(static_cast<int>(code)) + ": " + message; }
};

template <typename T = void>
class Result;

template <>
class Result<void> {
public:
    static Result ok() { return Result(); }
    static Result error(ErrorCode c, const std::string& m) { Result r; r.err_ = Error(c, m); return r; }
    static Result error(const Error& e) { Result r; r.err_ = e; return r; }
    bool isOk() const { return err_.isOk(); }
    bool hasError() const { return !err_.isOk(); }
    const Error& error() const { return err_; }
    explicit operator bool() const { return isOk(); }
private:
    Error err_;
};

template <typename T>
class Result {
public:
    static Result ok(T v) { Result r; r.val_ = std::move(v); return r; }
    static Result error(ErrorCode c, const std::string& m) { Result r; r.err_ = Error(c, m); return r; }
    static Result error(const Error& e) { Result r; r.err_ = e; return r; }
    bool isOk() const { return err_.isOk(); }
    bool hasError() const { return !err_.isOk(); }
    const Error& error() const { return err_; }
    const T& value() const { if (hasError()) throw std::runtime_error("Result contains an error, not a value"); return val_; }
    T& value() { if (hasError()) throw std::runtime_error("Result contains an error, not a value"); return val_; }
    explicit operator bool() const { return isOk(); }
private:
    T val_{};
    Error err_;
};

// ---- server/config.h (hot-path keys only) ----------------------------------------------------------
struct DetectionConfig {
    uint16_t model_width = 416;      // configs/server.json:30
    uint16_t model_height = 416;     // configs/server.json:31
};

struct ServerConfig {
    std::string model_path = "models/yolo_nano_cs16.onnx";
    std::string inference_engine = "onnx";
    uint32_t target_fps = 60;
    float confidence_threshold = 0.5f;
    float nms_threshold = 0.45f;
    size_t max_queue_size = 8;
    bool use_cpu_affinity = true;
    int cpu_core_id = 0;
    bool use_high_priority = true;
    uint8_t worker_threads = 2;
    DetectionConfig detection;
};

// ---- inference/inference_engine.h ------------------------------------------------------------------
struct InferenceRequest {
    uint32_t client_id = 0;
    uint32_t frame_id = 0;
    uint64_t timestamp = 0;
    uint16_t width = 0, height = 0;
    std::vector<uint8_t> data;       // width*height*3 bytes, BGR interleaved
    bool is_keyframe = false;
};

using InferenceCallback = std::function<void(uint32_t client_id, const GameState& state)>;

class IInferenceEngine {
public:
    virtual ~IInferenceEngine() = default;
    virtual Result<void> initialize() = 0;
    virtual Result<void> shutdown() = 0;
    virtual Result<void> submitInference(const InferenceRequest& request) = 0;
    virtual void setCallback(InferenceCallback callback) = 0;
    virtual size_t getQueueSize() const = 0;
    virtual std::string getName() const = 0;
    virtual std::unordered_map<std::string, std::string> getStatus() const = 0;
};

class IInferenceEngineFactory {
public:
    virtual ~IInferenceEngineFactory() = default;
    virtual std::unique_ptr<IInferenceEngine> createEngine(const ServerConfig& config) = 0;
    virtual std::string getName() const = 0;
};

class InferenceEngineManager {
public:
    static InferenceEngineManager& getInstance() { static InferenceEngineManager m; return m; }
    void registerFactory(std::shared_ptr<IInferenceEngineFactory> factory) {
        if (factory) factories_[factory->getName()] = std::move(factory);
    }
    std::unique_ptr<IInferenceEngine> createEngine(const std::string& name, const ServerConfig& config) {
        auto it = factories_.find(name);
        return it == factories_.end() ? nullptr : it->second->createEngine(config);
    }
    std::vector<std::string> getAvailableEngines() const {
        std::vector<std::string> names;
        for (const auto& kv : factories_) names.push_back(kv.first);
        return names;
    }
    bool isEngineAvailable(const std::string& name) const { return factories_.count(name) != 0; }
private:
    InferenceEngineManager() = default;
    InferenceEngineManager(const InferenceEngineManager&) = delete;
    InferenceEngineManager& operator=(const InferenceEngineManager&) = delete;
    std::map<std::string, std::shared_ptr<IInferenceEngineFactory>> factories_;
};

#define REGISTER_INFERENCE_ENGINE(factory_class)                                                      \
    namespace {                                                                                       \
    struct Register##factory_class {                                                                  \
        Register##factory_class() {                                                                   \
            zero_latency::InferenceEngineManager::getInstance().registerFactory(                      \
                std::make_shared<factory_class>());                                                   \
        }                                                                                             \
    };                                                                                                \
    static Register##factory_class register_##factory_class;                                          \
    }

}  // namespace zero_latency
