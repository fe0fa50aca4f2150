// zly_frame_server.hpp -- the glue NetworkServer provides around the engine, without the socket: packet bytes in ->
// InferenceRequest -> IInferenceEngine -> game-adapter step -> packet bytes out.  Restates
// NetworkServer::handleFrameData (reference src/network/network_server.cpp:184-241) and ::onInferenceResult (:243-283)
// on top of host/zly_wire.hpp and host/zly_game_step.hpp, so that a bytes-in / bytes-out harness around the plugin can
// be tested end to end.  The transport (ReliableUdp), client registry and event bus of the reference are out of scope.
#pragma once

#include "zly_game_step.hpp"
#include "zly_wire.hpp"

#include <atomic>
#include <chrono>
#include <mutex>

namespace zero_latency {

class FrameServer {
  public:
    using SendFn = std::function<void(uint32_t client_id, const std::vector<uint8_t>& packet)>;

    // max_frame_bytes: the largest chunked frame the server will reassemble (FrameAssembler allocates a frame's buffer when its first piece arrives);
    // pass a few times model_w * model_h * 3.  The default admits a 1920 x 1080 BGR frame.
    FrameServer(IInferenceEngine& engine, Cs16DetectionStep& adapter, SendFn send, size_t max_frame_bytes = (size_t)8 << 20)
        : engine_(engine), adapter_(adapter), send_(std::move(send)), assembler_(4, max_frame_bytes)
    {
        // NetworkServer's constructor wires the engine callback to onInferenceResult (network_server.cpp:21-22)
        engine_.setCallback([this](uint32_t client_id, const GameState& state) { onInferenceResult(client_id, state); });
    }

    // one datagram from `client_id` (the reference resolves the id from the sender address, :88-110): a whole frame (FrameDataPacket,
    // the reference's format: frames up to 65518 bytes) or one piece of a chunked raw frame (FrameChunkPacket, zly_wire.hpp)
    Result<void> onPacket(uint32_t client_id, const uint8_t* data, size_t size)
    {
        // nothing a datagram contains may take the server down: an allocation failure (or any other exception below) becomes an error result
        try { return onPacketImpl(client_id, data, size); }
        catch (const std::bad_alloc&) { ++bad_packets_; return Result<void>::error(ErrorCode::SYSTEM_ERROR, "out of memory while handling a packet"); }
        catch (const std::exception& ex) { ++bad_packets_; return Result<void>::error(ErrorCode::SYSTEM_ERROR, std::string("exception while handling a packet: ") + ex.what()); }
    }

    uint64_t reassembledFrames() const { return reassembled_frames_; }
    uint64_t droppedIncompleteFrames() const { std::lock_guard<std::mutex> lk(assembler_mutex_); return assembler_.dropped(); }
    uint64_t badPackets() const { return bad_packets_; }
    uint64_t sentPackets() const { return sent_packets_; }

  private:
    Result<void> onPacketImpl(uint32_t client_id, const uint8_t* data, size_t size)
    {
        if (size > 5 && data[5] == wire::kTypeFrameChunk) {
            auto chunk = wire::parseFrameChunk(data, size);
            if (chunk.hasError()) { ++bad_packets_; return Result<void>::error(chunk.error()); }
            wire::FrameData whole;
            Result<bool> done = Result<bool>::ok(false);
            {
                std::lock_guard<std::mutex> lk(assembler_mutex_);         // the reference's receive path is one thread; be safe for several
                done = assembler_.add(client_id, chunk.value(), &whole);
            }
            if (done.hasError()) { ++bad_packets_; return Result<void>::error(done.error()); }
            if (!done.value()) return Result<void>::ok();                 // more pieces to come
            ++reassembled_frames_;
            auto req = wire::frameToRequest(whole, client_id);
            if (req.hasError()) { ++bad_packets_; return Result<void>::error(req.error()); }
            return engine_.submitInference(req.value());
        }
        auto frame = wire::parseFrameData(data, size);
        if (frame.hasError()) { ++bad_packets_; return Result<void>::error(frame.error()); }
        auto req = wire::frameToRequest(frame.value(), client_id);
        if (req.hasError()) { ++bad_packets_; return Result<void>::error(req.error()); }
        return engine_.submitInference(req.value());
    }

    void onInferenceResult(uint32_t client_id, const GameState& state)
    {
        auto processed = adapter_.processDetections(client_id, state, Cs16DetectionStep::kGameCs16);
        if (processed.hasError()) return;                                  // :266-269: logged and dropped
        const uint64_t now = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        auto pkt = wire::serializeDetectionResult(processed.value(), sequence_++, now);
        if (pkt.hasError()) return;
        ++sent_packets_;
        send_(client_id, pkt.value());
    }

    IInferenceEngine& engine_;
    Cs16DetectionStep& adapter_;
    SendFn send_;
    std::atomic<uint32_t> sequence_{0};
    std::atomic<uint64_t> bad_packets_{0}, sent_packets_{0}, reassembled_frames_{0};
    mutable std::mutex assembler_mutex_;
    wire::FrameAssembler assembler_;
};

}  // namespace zero_latency
