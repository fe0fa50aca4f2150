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
static size_t kHeaderOffsetOfChunkOffset() { return 22 + 21; }
static void refinish(std::vector<uint8_t>& pkt)          // recompute the checksum of a hand-modified packet
{
    pkt[20] = 0; pkt[21] = 0;
    const uint16_t c = wire::crc16(pkt.data() + 2, pkt.size() - 2);
    std::memcpy(pkt.data() + 20, &c, 2);
}

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
    {   // chunked raw frames (extension): <dir>/chunks.bin = u32 count, then {u32 nbytes, packet} in ARRIVAL order (shuffled, with a
        // duplicate) made by the Python oracle -> reassembled frame -> <dir>/chunks_frame.bin; and the same frame cut here -> <dir>/chunks_out.bin
        std::vector<uint8_t> blob = slurp(dir + "/chunks.bin");
        if (blob.size() >= 4) {
            uint32_t n = 0; std::memcpy(&n, blob.data(), 4);
            size_t o = 4;
            wire::FrameAssembler asm_(2);
            wire::FrameData whole;
            int completed = 0, errors = 0, complete_at = -1;
            for (uint32_t i = 0; i < n && o + 4 <= blob.size(); ++i) {
                uint32_t nb = 0; std::memcpy(&nb, blob.data() + o, 4); o += 4;
                auto c = wire::parseFrameChunk(blob.data() + o, nb);
                if (c.hasError()) ++errors;
                else {
                    auto d = asm_.add(42, c.value(), &whole);
                    if (d.hasError()) ++errors;
                    else if (d.value()) { ++completed; complete_at = (int)i; }
                }
                o += nb;
            }
            rep << "chunks_completed=" << completed << "\nchunks_errors=" << errors << "\nchunks_complete_at=" << complete_at << "\nchunks_pending=" << asm_.pending(42) << "\n";
            if (completed) {
                rep << "chunks_frame_id=" << whole.frame_id << "\nchunks_ts=" << whole.timestamp << "\nchunks_w=" << whole.width << "\nchunks_h=" << whole.height
                    << "\nchunks_key=" << (whole.keyframe ? 1 : 0) << "\n";
                dump(dir + "/chunks_frame.bin", whole.data);
                auto cut = wire::serializeFrameChunks(whole, 60000, 700, 555555);
                rep << "chunks_serialize=" << static_cast<int>(cut.error().code) << "\n";
                if (cut.isOk()) {
                    std::vector<uint8_t> all;
                    for (const auto& pk : cut.value()) { const uint32_t nb = (uint32_t)pk.size(); all.insert(all.end(), (const uint8_t*)&nb, (const uint8_t*)&nb + 4); all.insert(all.end(), pk.begin(), pk.end()); }
                    dump(dir + "/chunks_out.bin", all);
                }
                // refused: a piece that runs past the frame, a piece that contradicts its frame, an index beyond the count
                auto one = wire::serializeFrameChunks(whole, 60000, 1, 1).value()[0];
                auto bad = one; bad[kHeaderOffsetOfChunkOffset()] = 0xFF; bad[kHeaderOffsetOfChunkOffset() + 1] = 0xFF; bad[kHeaderOffsetOfChunkOffset() + 2] = 0xFF;
                refinish(bad);
                rep << "chunk_past_end=" << static_cast<int>(wire::parseFrameChunk(bad.data(), bad.size()).error().code) << "\n";
                bad = one; bad[22 + 17] = 9; bad[22 + 18] = 0; bad[22 + 19] = 3; bad[22 + 20] = 0; refinish(bad);       // index 9 of 3
                rep << "chunk_bad_index=" << static_cast<int>(wire::parseFrameChunk(bad.data(), bad.size()).error().code) << "\n";
                wire::FrameAssembler a2(2);
                wire::FrameData tmp;
                auto c0 = wire::parseFrameChunk(one.data(), one.size());
                a2.add(1, c0.value(), &tmp);
                bad = one; bad[22 + 4] ^= 1; refinish(bad);                                                            // the same piece under another frame timestamp
                auto c1 = wire::parseFrameChunk(bad.data(), bad.size());
                rep << "chunk_contradicts=" << (c1.isOk() ? static_cast<int>(a2.add(1, c1.value(), &tmp).error().code) : -static_cast<int>(c1.error().code)) << "\n";
                // pieces must tile the frame uniformly -- piece i = bytes [i * P, ...): an overlapping piece (index 1 at offset 30000 behind a
                // 60000-byte piece 0) and a piece whose count does not match ceil(total / P) are refused by the parser itself, before anything is stored
                bad = one; bad[22 + 17] = 1; bad[22 + 21] = 0x30; bad[22 + 22] = 0x75; bad[22 + 23] = 0; bad[22 + 24] = 0; refinish(bad);   // index 1, offset 30000
                rep << "chunk_overlap=" << static_cast<int>(wire::parseFrameChunk(bad.data(), bad.size()).error().code) << "\n";
                bad = one; bad[22 + 19] = 7; refinish(bad);                                                            // 7 pieces claimed for a frame that needs 9
                rep << "chunk_bad_count=" << static_cast<int>(wire::parseFrameChunk(bad.data(), bad.size()).error().code) << "\n";
                // ... and ONE datagram cannot make the server allocate gigabytes: a self-consistent first piece of a 30000 x 30000 frame (2.7 GB, 45000 pieces)
                // parses, and the assembler refuses it (PACKET_TOO_LARGE) without allocating; nothing stays pending
                bad = one; bad[22 + 12] = 0x30; bad[22 + 13] = 0x75; bad[22 + 14] = 0x30; bad[22 + 15] = 0x75; bad[22 + 19] = 0xC8; bad[22 + 20] = 0xAF; refinish(bad);
                auto ch = wire::parseFrameChunk(bad.data(), bad.size());
                wire::FrameAssembler a4(4);
                rep << "chunk_huge_parse=" << (ch.isOk() ? 0 : static_cast<int>(ch.error().code)) << "\nchunk_huge_add=" << (ch.isOk() ? static_cast<int>(a4.add(3, ch.value(), &tmp).error().code) : -1)
                    << "\nchunk_huge_pending=" << a4.pending(3) << "\n";
                wire::FrameAssembler a5(4, 519168);                       // a limit of exactly one model-sized frame admits it
                rep << "chunk_limit_exact=" << (a5.add(3, c0.value(), &tmp).isOk() ? 1 : 0) << "\n";
                // eviction: a third incomplete frame of the same client pushes the oldest out
                wire::FrameAssembler a3(2);
                for (uint32_t id = 1; id <= 3; ++id) {
                    wire::FrameData f2 = whole; f2.frame_id = id;
                    auto pk = wire::serializeFrameChunks(f2, 60000, 1, 1).value()[0];
                    a3.add(5, wire::parseFrameChunk(pk.data(), pk.size()).value(), &tmp);
                }
                rep << "chunk_evicted=" << a3.dropped() << "\nchunk_pending_after=" << a3.pending(5) << "\n";
            }
        }
    }
    GameState big;                                                        // 1700 detections * 40 B > 65535: refused, not truncated
    big.detections.resize(1700);
    rep << "dets_too_big=" << static_cast<int>(wire::serializeDetectionResult(big, 1, 1).error().code) << "\n";
    return 0;
}
