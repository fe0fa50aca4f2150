// test_wire.cpp -- CPU-only driver for host/zly_wire.hpp.  usage: test_wire <dir>
//   reads  <dir>/frame.pkt            a FrameDataPacket made by the Python oracle
//   writes <dir>/dets.pkt             a DetectionResultPacket for a fixed GameState
//          <dir>/frame_roundtrip.pkt  the parsed frame serialised again (must equal frame.pkt byte for byte)
//          <dir>/report.txt           key=value lines: parsed fields and the error codes of the corrupted variants
#include "zly_wire.hpp"
#include "zly_sha256.hpp"

#include <cstdio>
#include <fstream>
#include <iterator>

using namespace zero_latency;

static std::vector<uint8_t> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {}); }
static void dump(const std::string& p, const std::vector<uint8_t>& v) { std::ofstream f(p, std::ios::binary); f.write((const char*)v.data(), (std::streamsize)v.size()); }

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    std::ofstream rep(dir + "/report.txt");
    std::vector<uint8_t> pkt = slurp(dir + "/frame.pkt");
    wire::Header h;
    auto fr = wire::parseFrameData(pkt.data(), pkt.size(), &h);
    rep << "parse=" << static_cast<int>(fr.error().code) << "\n";
    if (fr.isOk()) {
        const wire::FrameData& f = fr.value();
        rep << "frame_id=" << f.frame_id << "\ntimestamp=" << f.timestamp << "\nwidth=" << f.width << "\nheight=" << f.height
            << "\nkeyframe=" << (f.keyframe ? 1 : 0) << "\nbytes=" << f.data.size() << "\nsequence=" << h.sequence << "\npacket_ts=" << h.timestamp << "\n";
        auto again = wire::serializeFrameData(f, h.sequence, h.timestamp);
        if (again.isOk()) dump(dir + "/frame_roundtrip.pkt", again.value());
        auto req = wire::frameToRequest(f, 42);
        rep << "request=" << static_cast<int>(req.error().code) << "\n";
        if (req.isOk()) rep << "req_client=" << req.value().client_id << "\nreq_frame=" << req.value().frame_id << "\nreq_keyframe=" << (req.value().is_keyframe ? 1 : 0) << "\n";
        wire::FrameData bad = f;
        bad.data.pop_back();
        rep << "request_short=" << static_cast<int>(wire::frameToRequest(bad, 42).error().code) << "\n";
        bad.data.clear();
        rep << "request_empty=" << static_cast<int>(wire::frameToRequest(bad, 42).error().code) << "\n";
    }
    // corrupted variants of the same packet
    {
        auto c = pkt; c[30] ^= 0x01;                                   rep << "bad_crc=" << static_cast<int>(wire::parseFrameData(c.data(), c.size()).error().code) << "\n";
        c = pkt; c[0] ^= 0xFF;                                         rep << "bad_magic=" << static_cast<int>(wire::parseFrameData(c.data(), c.size()).error().code) << "\n";
        c = pkt; c.pop_back();                                         rep << "bad_length=" << static_cast<int>(wire::parseFrameData(c.data(), c.size()).error().code) << "\n";
        c = pkt;                                                       rep << "bad_type=" << static_cast<int>(wire::parseDetectionResult(c.data(), c.size()).error().code) << "\n";
        c.assign(pkt.begin(), pkt.begin() + 10);                       rep << "too_small=" << static_cast<int>(wire::parseFrameData(c.data(), c.size()).error().code) << "\n";
        // the first two bytes are NOT covered by the checksum (protocol.h:182-185): flipping them fails on the magic, not the CRC
    }
    GameState s;
    s.frame_id = 77; s.timestamp = 1234567890123ull;
    for (int i = 0; i < 3; ++i) {
        Detection d{};
        d.box = BoundingBox{0.1f * (i + 1), 0.2f, 0.3f, 0.4f};
        d.confidence = 0.5f + 0.125f * i; d.class_id = i * 7; d.track_id = 100 + i; d.timestamp = 999000 + i;
        s.detections.push_back(d);
    }
    auto out = wire::serializeDetectionResult(s, 5, 424242);
    rep << "dets_serialize=" << static_cast<int>(out.error().code) << "\n";
    if (out.isOk()) {
        dump(dir + "/dets.pkt", out.value());
        auto back = wire::parseDetectionResult(out.value().data(), out.value().size());
        rep << "dets_parse=" << static_cast<int>(back.error().code) << "\ndets_count=" << (back.isOk() ? back.value().detections.size() : 0) << "\n";
    }
    {   // SHA-256 of the model watch (FIPS 180-4 vectors; the file itself as a multi-block input)
        Sha256 a; a.update("abc", 3); rep << "sha_abc=" << a.hex() << "\n";
        Sha256 e; rep << "sha_empty=" << e.hex() << "\n";
        Sha256 m; const char* s56 = "abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq"; m.update(s56, 56); rep << "sha_56=" << m.hex() << "\n";
        rep << "sha_frame_pkt=" << sha256File(dir + "/frame.pkt") << "\nsha_missing=" << sha256File(dir + "/nope") << "\n";
    }
    GameState big;                                                        // 1700 detections * 40 B > 65535: refused, not truncated
    big.detections.resize(1700);
    rep << "dets_too_big=" << static_cast<int>(wire::serializeDetectionResult(big, 1, 1).error().code) << "\n";
    return 0;
}
