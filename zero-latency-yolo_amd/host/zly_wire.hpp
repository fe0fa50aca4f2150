// zly_wire.hpp -- the wire format either side of the detect path: what
// NetworkServer does between the UDP socket and IInferenceEngine (reference src/network/network_server.cpp:184-283),
// restated from the packet definitions in reference src/common/protocol.h.  Host-only C++, no GPU.
//
//   PacketHeader            22 bytes, packed: magic u32 @0 (0x59544C5A), version u8 @4 (1), type u8 @5,
//                           length u16 @6 (body bytes), sequence u32 @8, timestamp u64 @12 (ms), checksum u16 @20
//                           (protocol.h:38-73).  The reference's PROTOCOL_HEADER_SIZE = 16 constant is wrong;
//                           every code path uses sizeof(PacketHeader) = 22.
//   checksum                CRC-16-CCITT (poly 0x1021, init 0xFFFF, no reflection, no final xor; protocol.h:76-89)
//                           over bytes [2, size) of the packet with the checksum field zero -- i.e. it SKIPS the
//                           first two bytes of the magic although the checksum is the LAST header field
//                           (protocol.h:182-185, 242-245).  Reproduced as is: the bytes must match the reference's.
//   FrameDataPacket body    frame_id u32, timestamp u64, width u16, height u16, keyframe u8, then pixel bytes
//                           (protocol.h:407-440, 443-495); type 3.
//   DetectionResultPacket   frame_id u32, timestamp u64, count u16, count x Detection (40 raw bytes each)
//                           (protocol.h:541-567); type 4.
//   frame -> request        NetworkServer::handleFrameData (network_server.cpp:184-207): empty data / zero dims /
//                           data.size() != w*h*3 -> INVALID_INPUT (203); fields copied 1:1, client id from the session.
//
//   FrameChunkPacket        EXTENSION (no reference counterpart; type 8, the first value PacketType leaves free, types.h:73-82): a raw
//                           frame cut into pieces that fit the 16-bit length.  Same header / CRC rules; body = the FrameDataPacket
//                           fields (frame_id u32, timestamp u64, width u16, height u16, keyframe u8) + chunk_index u16,
//                           chunk_count u16, offset u32 (byte offset of this piece in the w*h*3 frame), then the piece.
//                           FrameAssembler puts the pieces of a frame back together (any order, duplicates ignored) and hands over
//                           the same FrameData a single FrameDataPacket would have carried.
//
// Known limits of the reference format, kept: `length` is 16 bits, so a body cannot exceed 65535 bytes (a raw
// 416x416 frame is 519168 bytes and does not fit one packet; the reference's client sends JPEG which its server
// cannot decode); FrameChunkPacket above is how a real frame reaches this server.  serialize* here refuse
// bodies that do not fit instead of truncating the length as the reference's static_cast does (protocol.h:178).
#pragma once

#include "zly_compat.hpp"

#include <cstring>
#include <deque>
#include <map>

namespace zero_latency {
namespace wire {

constexpr uint32_t kMagic = 0x59544C5A;
constexpr uint8_t kVersion = 1;
constexpr size_t kHeaderSize = 22;
constexpr uint8_t kTypeFrameData = 3;          // PacketType::FRAME_DATA (types.h:73-82)
constexpr uint8_t kTypeDetectionResult = 4;    // PacketType::DETECTION_RESULT
constexpr uint8_t kTypeFrameChunk = 8;         // extension: PacketType ends at CONFIG_UPDATE = 7
constexpr size_t kChunkBodyHeader = 25;        // frame_id 4 + timestamp 8 + width 2 + height 2 + keyframe 1 + index 2 + count 2 + offset 4
constexpr size_t kMaxChunkPayload = 0xFFFF - kChunkBodyHeader;

struct Header {
    uint32_t magic = kMagic;
    uint8_t version = kVersion, type = 0;
    uint16_t length = 0;
    uint32_t sequence = 0;
    uint64_t timestamp = 0;
    uint16_t checksum = 0;
};

inline uint16_t crc16(const uint8_t* data, size_t size)
{
    uint16_t crc = 0xFFFF;
    for (size_t i = 0; i < size; ++i) {
        crc = (uint16_t)(crc ^ ((uint16_t)data[i] << 8));
        for (int b = 0; b < 8; ++b) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
    }
    return crc;
}

namespace detail {
template <typename T> inline void put(std::vector<uint8_t>& b, size_t off, T v) { std::memcpy(b.data() + off, &v, sizeof v); }
template <typename T> inline T get(const uint8_t* p) { T v; std::memcpy(&v, p, sizeof v); return v; }

// header + body -> packet bytes, checksum last (protocol.h:150-193)
inline bool finish(std::vector<uint8_t>& pkt, uint8_t type, uint32_t sequence, uint64_t timestamp)
{
    const size_t body = pkt.size() - kHeaderSize;
    if (body > 0xFFFF) return false;
    put<uint32_t>(pkt, 0, kMagic); put<uint8_t>(pkt, 4, kVersion); put<uint8_t>(pkt, 5, type);
    put<uint16_t>(pkt, 6, (uint16_t)body); put<uint32_t>(pkt, 8, sequence); put<uint64_t>(pkt, 12, timestamp);
    put<uint16_t>(pkt, 20, 0);
    put<uint16_t>(pkt, 20, crc16(pkt.data() + 2, pkt.size() - 2));
    return true;
}
}  // namespace detail

// Packet::deserialize's header checks (protocol.h:201-258): size, magic/version, length, type, checksum.
inline Result<Header> parseHeader(const uint8_t* data, size_t size, uint8_t expected_type)
{
    using R = Result<Header>;
    if (size < kHeaderSize) return R::error(ErrorCode::INVALID_PACKET, "Packet too small");
    Header h;
    h.magic = detail::get<uint32_t>(data); h.version = data[4]; h.type = data[5];
    h.length = detail::get<uint16_t>(data + 6); h.sequence = detail::get<uint32_t>(data + 8);
    h.timestamp = detail::get<uint64_t>(data + 12); h.checksum = detail::get<uint16_t>(data + 20);
    if (h.magic != kMagic || h.version != kVersion) return R::error(ErrorCode::PROTOCOL_ERROR, "Invalid packet magic or version");
    if (kHeaderSize + h.length != size)
        return R::error(ErrorCode::INVALID_PACKET, "Invalid packet length: expected " + std::to_string(kHeaderSize + h.length) + ", got " + std::to_string(size));
    if (h.type != expected_type)
        return R::error(ErrorCode::PROTOCOL_ERROR, "Invalid packet type: expected " + std::to_string(expected_type) + ", got " + std::to_string(h.type));
    std::vector<uint8_t> tmp(data, data + size);
    tmp[20] = 0; tmp[21] = 0;
    const uint16_t crc = crc16(tmp.data() + 2, tmp.size() - 2);
    if (crc != h.checksum)
        return R::error(ErrorCode::PROTOCOL_ERROR, "Invalid packet checksum: expected " + std::to_string(h.checksum) + ", calculated " + std::to_string(crc));
    return R::ok(h);
}

struct FrameData {                     // common/types.h:28-34
    uint32_t frame_id = 0;
    uint64_t timestamp = 0;
    uint16_t width = 0, height = 0;
    std::vector<uint8_t> data;
    bool keyframe = false;
};

inline Result<std::vector<uint8_t>> serializeFrameData(const FrameData& f, uint32_t sequence, uint64_t packet_timestamp)
{
    std::vector<uint8_t> pkt(kHeaderSize + 17 + f.data.size());
    size_t o = kHeaderSize;
    detail::put<uint32_t>(pkt, o, f.frame_id); o += 4;
    detail::put<uint64_t>(pkt, o, f.timestamp); o += 8;
    detail::put<uint16_t>(pkt, o, f.width); o += 2;
    detail::put<uint16_t>(pkt, o, f.height); o += 2;
    pkt[o++] = f.keyframe ? 1 : 0;
    if (!f.data.empty()) std::memcpy(pkt.data() + o, f.data.data(), f.data.size());
    if (!detail::finish(pkt, kTypeFrameData, sequence, packet_timestamp))
        return Result<std::vector<uint8_t>>::error(ErrorCode::PACKET_TOO_LARGE, "frame does not fit the 16-bit packet length");
    return Result<std::vector<uint8_t>>::ok(std::move(pkt));
}

// FrameDataPacket::deserializeBody (protocol.h:443-495)
inline Result<FrameData> parseFrameData(const uint8_t* data, size_t size, Header* header_out = nullptr)
{
    using R = Result<FrameData>;
    auto h = parseHeader(data, size, kTypeFrameData);
    if (h.hasError()) return R::error(h.error());
    if (header_out) *header_out = h.value();
    const uint16_t length = h.value().length;
    if (length < 17)
        return R::error(ErrorCode::INVALID_PACKET, "Invalid frame data packet body length: expected at least 17, got " + std::to_string(length));
    const uint8_t* b = data + kHeaderSize;
    FrameData f;
    f.frame_id = detail::get<uint32_t>(b); f.timestamp = detail::get<uint64_t>(b + 4);
    f.width = detail::get<uint16_t>(b + 12); f.height = detail::get<uint16_t>(b + 14);
    f.keyframe = b[16] == 1;
    if (f.width == 0 || f.height == 0)
        return R::error(ErrorCode::INVALID_PACKET, "Invalid frame dimensions: " + std::to_string(f.width) + "x" + std::to_string(f.height));
    f.data.assign(b + 17, b + length);             // "non-strict": whatever follows is the pixel payload
    return R::ok(std::move(f));
}

// NetworkServer::handleFrameData (network_server.cpp:184-207): FrameData -> InferenceRequest, or INVALID_INPUT
inline Result<InferenceRequest> frameToRequest(const FrameData& f, uint32_t client_id)
{
    using R = Result<InferenceRequest>;
    if (f.data.empty() || f.width == 0 || f.height == 0) return R::error(ErrorCode::INVALID_INPUT, "Invalid frame data");
    const size_t expected = (size_t)f.width * f.height * 3;
    if (f.data.size() != expected)
        return R::error(ErrorCode::INVALID_INPUT, "Frame data size mismatch: expected " + std::to_string(expected) + " bytes, but received " + std::to_string(f.data.size()) + " bytes");
    InferenceRequest r;
    r.client_id = client_id; r.frame_id = f.frame_id; r.timestamp = f.timestamp;
    r.width = f.width; r.height = f.height; r.data = f.data; r.is_keyframe = f.keyframe;
    return R::ok(std::move(r));
}

// DetectionResultPacket::serializeBody (protocol.h:541-567): what onInferenceResult sends back (network_server.cpp:266-277)
inline Result<std::vector<uint8_t>> serializeDetectionResult(const GameState& s, uint32_t sequence, uint64_t packet_timestamp)
{
    std::vector<uint8_t> pkt(kHeaderSize + 14 + s.detections.size() * sizeof(Detection));
    size_t o = kHeaderSize;
    detail::put<uint32_t>(pkt, o, s.frame_id); o += 4;
    detail::put<uint64_t>(pkt, o, s.timestamp); o += 8;
    detail::put<uint16_t>(pkt, o, (uint16_t)s.detections.size()); o += 2;
    for (const Detection& d : s.detections) { std::memcpy(pkt.data() + o, &d, sizeof(Detection)); o += sizeof(Detection); }
    if (!detail::finish(pkt, kTypeDetectionResult, sequence, packet_timestamp))
        return Result<std::vector<uint8_t>>::error(ErrorCode::PACKET_TOO_LARGE, "detections do not fit the 16-bit packet length");
    return Result<std::vector<uint8_t>>::ok(std::move(pkt));
}

inline Result<GameState> parseDetectionResult(const uint8_t* data, size_t size, Header* header_out = nullptr)
{
    using R = Result<GameState>;
    auto h = parseHeader(data, size, kTypeDetectionResult);
    if (h.hasError()) return R::error(h.error());
    if (header_out) *header_out = h.value();
    const uint16_t length = h.value().length;
    if (length < 14)
        return R::error(ErrorCode::INVALID_PACKET, "Invalid detection result packet body length: expected at least 14, got " + std::to_string(length));
    const uint8_t* b = data + kHeaderSize;
    GameState s;
    s.frame_id = detail::get<uint32_t>(b); s.timestamp = detail::get<uint64_t>(b + 4);
    const uint16_t count = detail::get<uint16_t>(b + 12);
    if (14 + (size_t)count * sizeof(Detection) > length)
        return R::error(ErrorCode::INVALID_PACKET, "Invalid detection count: expected space for " + std::to_string(count) +
                                                                " detections, but only have " + std::to_string((length - 14) / sizeof(Detection)));
    s.detections.resize(count);
    if (count) std::memcpy(s.detections.data(), b + 14, (size_t)count * sizeof(Detection));
    return R::ok(std::move(s));
}

// ---- chunked raw frames (extension) ---------------------------------------------------------------------------------------------
struct FrameChunk {
    uint32_t frame_id = 0;
    uint64_t timestamp = 0;
    uint16_t width = 0, height = 0;
    bool keyframe = false;
    uint16_t index = 0, count = 0;
    uint32_t offset = 0;
    const uint8_t* payload = nullptr;          // points into the parsed packet
    size_t payload_size = 0;
    uint32_t piece = 0;                        // bytes per piece of this frame's tiling (derived and checked by parseFrameChunk)
};

// a frame -> ceil(bytes / max_payload) packets, sequence numbers sequence0, sequence0 + 1, ...
inline Result<std::vector<std::vector<uint8_t>>> serializeFrameChunks(const FrameData& f, size_t max_payload, uint32_t sequence0, uint64_t packet_timestamp)
{
    using R = Result<std::vector<std::vector<uint8_t>>>;
    if (max_payload == 0 || max_payload > kMaxChunkPayload) return R::error(ErrorCode::INVALID_ARGUMENT, "chunk payload must be 1.." + std::to_string(kMaxChunkPayload) + " bytes");
    if (f.data.empty() || f.data.size() != (size_t)f.width * f.height * 3) return R::error(ErrorCode::INVALID_INPUT, "a chunked frame is raw w*h*3 bytes");
    const size_t count = (f.data.size() + max_payload - 1) / max_payload;
    if (count > 0xFFFF) return R::error(ErrorCode::PACKET_TOO_LARGE, "frame needs more than 65535 chunks");
    std::vector<std::vector<uint8_t>> out;
    for (size_t i = 0; i < count; ++i) {
        const size_t off = i * max_payload, nb = std::min(max_payload, f.data.size() - off);
        std::vector<uint8_t> pkt(kHeaderSize + kChunkBodyHeader + nb);
        size_t o = kHeaderSize;
        detail::put<uint32_t>(pkt, o, f.frame_id); o += 4;
        detail::put<uint64_t>(pkt, o, f.timestamp); o += 8;
        detail::put<uint16_t>(pkt, o, f.width); o += 2;
        detail::put<uint16_t>(pkt, o, f.height); o += 2;
        pkt[o++] = f.keyframe ? 1 : 0;
        detail::put<uint16_t>(pkt, o, (uint16_t)i); o += 2;
        detail::put<uint16_t>(pkt, o, (uint16_t)count); o += 2;
        detail::put<uint32_t>(pkt, o, (uint32_t)off); o += 4;
        std::memcpy(pkt.data() + o, f.data.data() + off, nb);
        if (!detail::finish(pkt, kTypeFrameChunk, sequence0 + (uint32_t)i, packet_timestamp)) return R::error(ErrorCode::PACKET_TOO_LARGE, "chunk does not fit the 16-bit packet length");
        out.push_back(std::move(pkt));
    }
    return R::ok(std::move(out));
}

inline Result<FrameChunk> parseFrameChunk(const uint8_t* data, size_t size, Header* header_out = nullptr)
{
    using R = Result<FrameChunk>;
    auto h = parseHeader(data, size, kTypeFrameChunk);
    if (h.hasError()) return R::error(h.error());
    if (header_out) *header_out = h.value();
    const uint16_t length = h.value().length;
    if (length < kChunkBodyHeader)
        return R::error(ErrorCode::INVALID_PACKET, "Invalid frame chunk packet body length: expected at least " + std::to_string(kChunkBodyHeader) + ", got " + std::to_string(length));
    const uint8_t* b = data + kHeaderSize;
    FrameChunk c;
    c.frame_id = detail::get<uint32_t>(b); c.timestamp = detail::get<uint64_t>(b + 4);
    c.width = detail::get<uint16_t>(b + 12); c.height = detail::get<uint16_t>(b + 14);
    c.keyframe = b[16] == 1;
    c.index = detail::get<uint16_t>(b + 17); c.count = detail::get<uint16_t>(b + 19); c.offset = detail::get<uint32_t>(b + 21);
    c.payload = b + kChunkBodyHeader; c.payload_size = (size_t)length - kChunkBodyHeader;
    if (c.width == 0 || c.height == 0)
        return R::error(ErrorCode::INVALID_PACKET, "Invalid frame dimensions: " + std::to_string(c.width) + "x" + std::to_string(c.height));
    const size_t total = (size_t)c.width * c.height * 3;
    auto bad = [&]() {
        return R::error(ErrorCode::INVALID_PACKET, "Invalid frame chunk: index " + std::to_string(c.index) + " of " + std::to_string(c.count) + ", bytes [" +
                                                           std::to_string(c.offset) + ", " + std::to_string((size_t)c.offset + c.payload_size) + ") of " + std::to_string(total));
    };
    if (c.count == 0 || c.index >= c.count || c.payload_size == 0 || (size_t)c.offset + c.payload_size > total) return bad();
    // The pieces of a frame tile it UNIFORMLY (serializeFrameChunks: piece i = bytes [i * P, min((i + 1) * P, total))), and every piece says enough to
    // check that on its own: P = offset / index (piece 0: its payload size; a single piece: the frame).  A piece that does not fit the tiling is refused
    // here, so pieces that pass cannot overlap or leave holes, whatever order they arrive in (the assembler used to accept overlapping
    // pieces whose sizes happened to add up).
    size_t P = 0;
    if (c.count == 1) P = total;
    else if (c.index == 0) P = c.payload_size;
    else { if (c.offset % c.index) return bad(); P = c.offset / c.index; }
    if (P == 0 || P > kMaxChunkPayload || (total + P - 1) / P != c.count || (size_t)c.index * P != c.offset || c.payload_size != std::min(P, total - c.offset)) return bad();
    c.piece = (uint32_t)P;
    return R::ok(c);
}

// Reassembly per (client, frame_id).  Pieces may arrive in any order and more than once; a piece that contradicts the frame's first
// piece (dimensions, count, timestamp) is refused.  At most `max_pending` incomplete frames are kept per client: UDP loses datagrams, and
// a frame that lost one never completes -- the oldest incomplete frame makes room (counted in dropped()).
class FrameAssembler {
  public:
    // max_frame_bytes bounds what ONE datagram can make the server allocate: a piece's header claims the frame's size (w, h up to 65535: 12.9 GB),
    // and the buffer is allocated when the first piece arrives.  Default 8 MiB (a 1920 x 1080 BGR frame is 6.2 MB); FrameServer passes a few times
    // the model's frame size.  With max_pending frames per client that is the most a client can hold.
    explicit FrameAssembler(size_t max_pending = 4, size_t max_frame_bytes = (size_t)8 << 20)
        : max_pending_(max_pending ? max_pending : 1), max_frame_bytes_(max_frame_bytes) {}

    // -> true when this piece completed its frame (*out then holds it), false when more pieces are needed
    Result<bool> add(uint32_t client_id, const FrameChunk& c, FrameData* out)
    {
        using R = Result<bool>;
        const size_t total = (size_t)c.width * c.height * 3;
        if (total > max_frame_bytes_)
            return R::error(ErrorCode::PACKET_TOO_LARGE, "chunked frame of " + std::to_string(total) + " bytes exceeds the " + std::to_string(max_frame_bytes_) + "-byte limit");
        // parseFrameChunk has checked the piece against the uniform tiling its own fields imply; a hand-built FrameChunk gets the same check here
        if (c.piece == 0 || c.count == 0 || c.index >= c.count || (total + c.piece - 1) / c.piece != c.count || (size_t)c.index * c.piece != c.offset ||
            c.payload_size != std::min((size_t)c.piece, total - std::min(total, (size_t)c.offset)) || c.payload_size == 0)
            return R::error(ErrorCode::INVALID_PACKET, "frame chunk does not fit the frame's tiling");
        std::deque<Partial>& q = pending_[client_id];
        Partial* p = nullptr;
        for (Partial& x : q) if (x.f.frame_id == c.frame_id) { p = &x; break; }
        if (!p) {
            if (q.size() >= max_pending_) { q.pop_front(); ++dropped_; }
            q.emplace_back();
            p = &q.back();
            p->f.frame_id = c.frame_id; p->f.timestamp = c.timestamp; p->f.width = c.width; p->f.height = c.height; p->f.keyframe = c.keyframe;
            p->piece = c.piece;
            p->f.data.assign(total, 0);
            p->have.assign(c.count, false);
        } else if (p->f.width != c.width || p->f.height != c.height || p->have.size() != c.count || p->f.timestamp != c.timestamp || p->piece != c.piece) {
            return R::error(ErrorCode::INVALID_PACKET, "frame chunk contradicts the earlier pieces of frame " + std::to_string(c.frame_id));
        }
        if (p->have[c.index]) return R::ok(false);                       // duplicate datagram
        std::memcpy(p->f.data.data() + c.offset, c.payload, c.payload_size);
        p->have[c.index] = true;
        if (++p->got < p->have.size()) return R::ok(false);              // every piece sits at index * piece and has the tiling's size: all indices seen = the frame is whole
        *out = std::move(p->f);
        erase(q, c.frame_id);
        return R::ok(true);
    }
    uint64_t dropped() const { return dropped_; }
    size_t pending(uint32_t client_id) const { auto it = pending_.find(client_id); return it == pending_.end() ? 0 : it->second.size(); }

  private:
    struct Partial { FrameData f; std::vector<bool> have; size_t got = 0; uint32_t piece = 0; };
    static void erase(std::deque<Partial>& q, uint32_t frame_id)
    {
        for (auto it = q.begin(); it != q.end(); ++it) if (it->f.frame_id == frame_id) { q.erase(it); return; }
    }
    size_t max_pending_, max_frame_bytes_;
    std::map<uint32_t, std::deque<Partial>> pending_;
    uint64_t dropped_ = 0;
};

}  // namespace wire
}  // namespace zero_latency
