// test_frame_server.cpp -- bytes in, bytes out around the plugin: FrameDataPacket datagrams -> FrameServer -> HipInferenceEngine
// -> game-adapter step -> DetectionResultPacket datagrams.   usage: test_frame_server <weights> <packets.bin> <out.bin>
// packets.bin: u32 count, then {u32 client, u32 nbytes, bytes}.  out.bin: u32 count, then {u32 client, u32 nbytes, bytes}
// in send order, followed by u64 bad_packets, u64 reassembled frames.
#include "zly_frame_server.hpp"

#include <condition_variable>
#include <cstdio>
#include <fstream>
#include <mutex>

using namespace zero_latency;

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    std::ifstream in(argv[2], std::ios::binary);
    uint32_t n = 0;
    in.read(reinterpret_cast<char*>(&n), 4);
    struct Pkt { uint32_t client; std::vector<uint8_t> bytes; };
    std::vector<Pkt> pkts(n);
    for (Pkt& p : pkts) {
        uint32_t nb = 0;
        in.read(reinterpret_cast<char*>(&p.client), 4);
        in.read(reinterpret_cast<char*>(&nb), 4);
        p.bytes.resize(nb);
        in.read(reinterpret_cast<char*>(p.bytes.data()), nb);
    }
    if (!in) return 2;

    ServerConfig config;
    config.model_path = argv[1];
    config.inference_engine = "hip";
    config.confidence_threshold = 0.02f;
    config.detection.model_width = 128;
    config.detection.model_height = 96;
    auto engine = InferenceEngineManager::getInstance().createEngine("hip", config);
    if (!engine) return 3;
    Cs16DetectionStep adapter;
    adapter.initialize();
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Pkt> sent;
    FrameServer server(*engine, adapter, [&](uint32_t client, const std::vector<uint8_t>& bytes) {
        std::lock_guard<std::mutex> lk(mu);
        sent.push_back(Pkt{client, bytes});
        cv.notify_all();
    });
    auto init = engine->initialize();
    if (init.hasError()) { std::fprintf(stderr, "%s\n", init.error().toString().c_str()); return 4; }
    size_t accepted = 0;
    for (const Pkt& p : pkts) {
        const bool piece = p.bytes.size() > 5 && p.bytes[5] == wire::kTypeFrameChunk;      // a piece of a chunked frame: a result only when its frame completes
        if (server.onPacket(p.client, p.bytes.data(), p.bytes.size()).isOk() && !piece) ++accepted;
    }
    accepted += (size_t)server.reassembledFrames();
    {
        std::unique_lock<std::mutex> lk(mu);
        if (!cv.wait_for(lk, std::chrono::seconds(60), [&] { return sent.size() >= accepted; })) return 6;
    }
    engine->shutdown();
    std::ofstream out(argv[3], std::ios::binary);
    const uint32_t m = (uint32_t)sent.size();
    out.write(reinterpret_cast<const char*>(&m), 4);
    for (const Pkt& p : sent) {
        const uint32_t nb = (uint32_t)p.bytes.size();
        out.write(reinterpret_cast<const char*>(&p.client), 4);
        out.write(reinterpret_cast<const char*>(&nb), 4);
        out.write(reinterpret_cast<const char*>(p.bytes.data()), nb);
    }
    const uint64_t bad = server.badPackets();
    out.write(reinterpret_cast<const char*>(&bad), 8);
    const uint64_t re = server.reassembledFrames();
    out.write(reinterpret_cast<const char*>(&re), 8);
    return 0;
}
