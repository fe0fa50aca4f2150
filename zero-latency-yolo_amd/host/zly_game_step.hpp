// zly_game_step.hpp -- the post-step between IInferenceEngine's callback and the DetectionResultPacket
// (SURVEY.md section 8f, rank 2): CS16GameAdapter::processDetections, reference
// src/game/games/cs16/cs16_game_adapter.cpp:36-69 and ::processCS16Detections :243-262, with the per-client
// tracked-object table of src/game/base/game_adapter_base.h:64-115.  Host-only C++: a few dozen bytes per frame.
//
//   for each detection, in order:   track_id == 0 -> track_id = next_track_id++   (one counter for ALL clients, starts at 1)
//                                   class_id == 2 (HEAD, constants.h:39) -> box.height *= head_size_factor (0.7, server.json:42-46)
//   then, per client:               tracked[track_id] = detection (last one wins), and every tracked object with
//                                   state.timestamp - object.timestamp > 100 (UNSIGNED 64-bit ms, so an object stamped
//                                   later than the frame also expires) is dropped.
// Errors as the reference: NOT_INITIALIZED (3) before initialize(), INVALID_ARGUMENT (2) when game id != CS_1_6 (1).
#pragma once

#include "zly_compat.hpp"

#include <mutex>

namespace zero_latency {

class Cs16DetectionStep {
  public:
    static constexpr uint8_t kGameCs16 = 1;     // GameType::CS_1_6 (types.h:107-113)
    static constexpr int32_t kClassHead = 2;    // constants::cs16::CLASS_HEAD

    explicit Cs16DetectionStep(float head_size_factor = 0.7f) : head_size_factor_(head_size_factor) {}
    void initialize() { initialized_ = true; }

    Result<GameState> processDetections(uint32_t client_id, const GameState& raw, uint8_t game_id)
    {
        using R = Result<GameState>;
        if (!initialized_) return R::error(ErrorCode::NOT_INITIALIZED, "Game adapter not initialized");
        if (game_id != kGameCs16) return R::error(ErrorCode::INVALID_ARGUMENT, "Unsupported game ID for CS16GameAdapter");
        std::lock_guard<std::mutex> lock(mutex_);
        GameState out = raw;
        for (Detection& d : out.detections) {
            if (d.track_id == 0) d.track_id = next_track_id_++;
            if (d.class_id == kClassHead) d.box.height *= head_size_factor_;
        }
        auto& tracked = clients_[client_id];
        for (const Detection& d : out.detections) tracked[d.track_id] = d;
        for (auto it = tracked.begin(); it != tracked.end();) {
            if ((uint64_t)(out.timestamp - it->second.timestamp) > 100) it = tracked.erase(it);
            else ++it;
        }
        return R::ok(std::move(out));
    }

    size_t trackedCount(uint32_t client_id) const
    {
        std::lock_guard<std::mutex> lock(mutex_);
        auto it = clients_.find(client_id);
        return it == clients_.end() ? 0 : it->second.size();
    }
    uint32_t nextTrackId() const { std::lock_guard<std::mutex> lock(mutex_); return next_track_id_; }

  private:
    mutable std::mutex mutex_;
    bool initialized_ = false;
    float head_size_factor_;
    uint32_t next_track_id_ = 1;
    std::unordered_map<uint32_t, std::unordered_map<uint32_t, Detection>> clients_;
};

}  // namespace zero_latency
