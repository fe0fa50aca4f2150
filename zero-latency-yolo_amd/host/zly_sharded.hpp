// zly_sharded.hpp -- ONE server process driving N MI355X: the product caller of libzly_gather.so (include/zly_gather.h).
//
// north-star / SURVEY.md section 8e: "incoming client frames are batched and sharded one-frame-per-GPU across the node with a trivial RCCL gather of
// detections over xGMI".  The reference has no counterpart (OnnxInferenceEngine runs one frame at a time on one device, reference
// src/inference/onnx_engine.cpp:518-646; its "dynamic batching" is a TODO, :348-365); what is kept of it is the request / result records
// (InferenceRequest, GameState: src/inference/inference_engine.h:16-31) and their semantics (frame_id / timestamp echo the request, :520-521).
//
//   detectBatch(requests):  frame i -> device i % N, slot i / N                                (shard.py's partition)
//     per device d:         frames -> pinned staging -> H2D on stream d -> zly_detect_device (preProcess .. applyNMS on the shard, result slabs stay in HBM)
//                           zly_join(stream d)                                                  (orders stream d behind the engine's NMS)
//     zly_gather_all:       ONE grouped ncclAllGather of the fixed-size slabs over xGMI        (the path's only exchange step; 64 x 2.6 KB per device)
//     device 0:             ONE device-to-host copy of the gathered slabs -> GameStates in global frame order
// The plugin's pipelined mode (HipInferenceEngine with ZLY_NUM_DEVICES) lets every GPU download its own slabs, which is right for independent client
// streams; this class is the lock-step form for a host that already holds a global batch (the frame server fed by many clients, the sharded leg of
// tools/bench_sharded.cpp).
//
// `Dev` supplies the few device-runtime calls (HIP in the product: zly_sharded_hip.hpp; a host-memory stand-in in tests/cpp/test_sharded_stub.cpp, which
// also stubs the two C ABIs -- test infrastructure, not a fallback: libzly.so has none).
#pragma once

#include "zly.h"
#include "zly_compat.hpp"
#include "zly_gather.h"

#include <cstring>
#include <string>
#include <vector>

namespace zero_latency {

template <class Dev>
class ShardedDetectorT {
  public:
    ShardedDetectorT() = default;
    ~ShardedDetectorT() { shutdown(); }
    ShardedDetectorT(const ShardedDetectorT&) = delete;
    ShardedDetectorT& operator=(const ShardedDetectorT&) = delete;

    // one engine per device first_device .. first_device + ndev - 1, each for up to `per_device` frames of model size per call
    Result<void> initialize(const ServerConfig& config, int ndev, int first_device, int per_device, int max_dets)
    {
        if (!devs_.empty()) return Result<void>::ok();
        if (ndev < 1 || per_device < 1 || max_dets < 1 || first_device < 0) return Result<void>::error(ErrorCode::INVALID_ARGUMENT, "bad sharding arguments");
        per_ = per_device; cap_ = max_dets;
        w_ = config.detection.model_width; h_ = config.detection.model_height;
        frame_bytes_ = (size_t)w_ * h_ * 3;
        std::vector<int32_t> ids;
        for (int d = 0; d < ndev; ++d) {
            PerDev pd;
            pd.device = first_device + d;
            zly_config c;
            zly_default_config(&c);
            c.weights_path = config.model_path.c_str();
            c.model_w = w_; c.model_h = h_;
            c.conf_thr = config.confidence_threshold; c.iou_thr = config.nms_threshold;
            c.max_batch = per_; c.max_dets = cap_; c.device = pd.device;
            c.flags = ZLY_FLAG_NO_HEAD_TENSOR;
            const int32_t rc = zly_create(&c, &pd.engine);
            if (rc != ZLY_OK) { const std::string m = zly_last_error(); devs_.push_back(pd); shutdown(); return Result<void>::error(static_cast<ErrorCode>(rc), "Failed to initialize HIP inference engine: " + m); }
            slab_bytes_ = zly_slab_bytes(pd.engine);
            bool ok = Dev::setDevice(pd.device) && Dev::streamCreate(&pd.stream) && Dev::hostAlloc((void**)&pd.h_frames, (size_t)per_ * frame_bytes_) &&
                      Dev::deviceAlloc(&pd.d_frames, (size_t)per_ * frame_bytes_) && Dev::deviceAlloc(&pd.d_slabs, (size_t)per_ * slab_bytes_) &&
                      Dev::deviceAlloc(&pd.d_all, (size_t)ndev * per_ * slab_bytes_);
            devs_.push_back(pd);
            ids.push_back(pd.device);
            if (!ok) { shutdown(); return Result<void>::error(ErrorCode::SYSTEM_ERROR, "device allocation failed on device " + std::to_string(pd.device)); }
        }
        if (!Dev::setDevice(devs_[0].device) || !Dev::hostAlloc((void**)&h_all_, (size_t)ndev * per_ * slab_bytes_)) { shutdown(); return Result<void>::error(ErrorCode::SYSTEM_ERROR, "host allocation failed"); }
        if (zly_gather_create(ndev, ids.data(), &gather_) != 0) { const std::string m = zly_gather_last_error(); shutdown(); return Result<void>::error(ErrorCode::SYSTEM_ERROR, "zly_gather_create: " + m); }
        return Result<void>::ok();
    }

    // up to ndev * per_device requests of model size (the lock-step path takes one frame size per call: zly_detect_device); results in request order
    Result<std::vector<GameState>> detectBatch(const std::vector<InferenceRequest>& reqs)
    {
        using R = Result<std::vector<GameState>>;
        const size_t N = devs_.size();
        if (N == 0 || !gather_) return R::error(ErrorCode::NOT_INITIALIZED, "Engine not running");
        if (reqs.empty() || reqs.size() > N * (size_t)per_) return R::error(ErrorCode::INVALID_ARGUMENT, "batch of " + std::to_string(reqs.size()) + " frames, capacity " + std::to_string(N * (size_t)per_));
        for (const InferenceRequest& r : reqs)
            if (r.width != w_ || r.height != h_ || r.data.size() != frame_bytes_)
                return R::error(ErrorCode::INVALID_INPUT, "Invalid image data size: expected " + std::to_string(frame_bytes_) + ", got " + std::to_string(r.data.size()));       // onnx_engine.cpp:659-665
        std::vector<const void*> send(N);
        std::vector<void*> recv(N), streams(N);
        for (size_t d = 0; d < N; ++d) {
            PerDev& pd = devs_[d];
            const size_t count = reqs.size() > d ? (reqs.size() - d + N - 1) / N : 0;          // frames d, d + N, d + 2N, ...
            if (!Dev::setDevice(pd.device)) return R::error(ErrorCode::SYSTEM_ERROR, "cannot select device");
            for (size_t slot = 0; slot < count; ++slot) std::memcpy(pd.h_frames + slot * frame_bytes_, reqs[slot * N + d].data.data(), frame_bytes_);
            bool ok = Dev::memsetAsync(pd.d_slabs, 0, (size_t)per_ * slab_bytes_, pd.stream);    // slots without a frame gather as empty slabs (n_kept = 0)
            if (ok && count) ok = Dev::copyH2DAsync(pd.d_frames, pd.h_frames, count * frame_bytes_, pd.stream);
            if (!ok) return R::error(ErrorCode::SYSTEM_ERROR, "upload failed on device " + std::to_string(pd.device));
            if (count) {
                int32_t rc = zly_detect_device(pd.engine, (int32_t)count, pd.d_frames, w_, h_, pd.d_slabs, (uint32_t)(step_ * 65536u), pd.stream);
                if (rc == ZLY_OK) rc = zly_join(pd.engine, pd.stream, 0);
                if (rc != ZLY_OK) return R::error(static_cast<ErrorCode>(rc), std::string("detect failed: ") + zly_last_error());
            }
            send[d] = pd.d_slabs; recv[d] = pd.d_all; streams[d] = pd.stream;
        }
        if (zly_gather_all(gather_, send.data(), recv.data(), (size_t)per_ * slab_bytes_, streams.data()) != 0)
            return R::error(ErrorCode::SYSTEM_ERROR, std::string("zly_gather_all: ") + zly_gather_last_error());
        const size_t all_bytes = N * (size_t)per_ * slab_bytes_;
        if (!Dev::setDevice(devs_[0].device) || !Dev::copyD2HAsync(h_all_, devs_[0].d_all, all_bytes, devs_[0].stream)) return R::error(ErrorCode::SYSTEM_ERROR, "download failed");
        for (size_t d = 0; d < N; ++d)                                                            // every stream: its staging buffers are reused by the next call
            if (!Dev::setDevice(devs_[d].device) || !Dev::streamSynchronize(devs_[d].stream)) return R::error(ErrorCode::INFERENCE_ERROR, "device fault on device " + std::to_string(devs_[d].device));
        ++step_;
        std::vector<GameState> out(reqs.size());
        for (size_t i = 0; i < reqs.size(); ++i) {
            const unsigned char* slab = h_all_ + ((i % N) * (size_t)per_ + i / N) * slab_bytes_;    // rank-major gather: rank i % N, slot i / N
            zly_slab_header hd;
            std::memcpy(&hd, slab, sizeof hd);
            const size_t n = (size_t)std::min<int32_t>(hd.n_kept, cap_);
            out[i].frame_id = reqs[i].frame_id; out[i].timestamp = reqs[i].timestamp;             // onnx_engine.cpp:520-521
            out[i].detections.resize(n);
            static_assert(sizeof(zly_det) == sizeof(Detection), "zly_det must be layout-identical to Detection");
            if (n) std::memcpy(out[i].detections.data(), slab + sizeof hd, n * sizeof(Detection));
        }
        return R::ok(std::move(out));
    }

    void shutdown()
    {
        if (gather_) { zly_gather_destroy(gather_); gather_ = nullptr; }
        for (PerDev& pd : devs_) {
            Dev::setDevice(pd.device);
            if (pd.stream) { Dev::streamSynchronize(pd.stream); Dev::streamDestroy(pd.stream); }
            if (pd.h_frames) Dev::hostFree(pd.h_frames);
            if (pd.d_frames) Dev::deviceFree(pd.d_frames);
            if (pd.d_slabs) Dev::deviceFree(pd.d_slabs);
            if (pd.d_all) Dev::deviceFree(pd.d_all);
            if (pd.engine) zly_destroy(pd.engine);
        }
        devs_.clear();
        if (h_all_) { Dev::hostFree(h_all_); h_all_ = nullptr; }
    }

    int devices() const { return (int)devs_.size(); }
    int capacity() const { return (int)devs_.size() * per_; }
    size_t gatheredBytesPerStep() const { return devs_.size() * (size_t)per_ * slab_bytes_; }

  private:
    struct PerDev {
        int device = 0;
        zly_engine* engine = nullptr;
        void* stream = nullptr;
        unsigned char* h_frames = nullptr;
        void* d_frames = nullptr; void* d_slabs = nullptr; void* d_all = nullptr;
    };
    std::vector<PerDev> devs_;
    zly_gather* gather_ = nullptr;
    unsigned char* h_all_ = nullptr;
    int per_ = 0, cap_ = 0, w_ = 0, h_ = 0;
    size_t frame_bytes_ = 0, slab_bytes_ = 0;
    uint32_t step_ = 0;
};

}  // namespace zero_latency
