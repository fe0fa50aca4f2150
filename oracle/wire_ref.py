"""CPU ORACLE (test infrastructure) for the wire format either side of the detect path: a plain-Python restatement
of reference src/common/protocol.h -- PacketHeader (:38-73), calculateCRC16 (:76-89), Packet::serialize (:150-193) and
::deserialize (:201-258), FrameDataPacket body (:407-440), DetectionResultPacket body (:541-567) -- and of
NetworkServer::handleFrameData's checks (src/network/network_server.cpp:184-207).

Pinned by known-answer tests derived from the cited lines (tests/test_wire.py): the reference ships no packet fixtures.
The CRC is CRC-16/CCITT-FALSE (check value 0x29B1 for b"123456789"), but it is applied to bytes [2:] of the packet."""
import struct

MAGIC, VERSION, HEADER = 0x59544C5A, 1, 22
FRAME_DATA, DETECTION_RESULT = 3, 4
FRAME_CHUNK = 8          # extension (host/zly_wire.hpp): the first value PacketType leaves free (types.h:73-82)


def crc16(data: bytes) -> int:
    crc = 0xFFFF
    for byte in data:
        crc ^= byte << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc


def packet(ptype: int, sequence: int, timestamp: int, body: bytes) -> bytes:
    assert len(body) <= 0xFFFF
    hdr = struct.pack("<IBBHIQH", MAGIC, VERSION, ptype, len(body), sequence, timestamp, 0)
    raw = bytearray(hdr + body)
    struct.pack_into("<H", raw, 20, crc16(bytes(raw[2:])))
    return bytes(raw)


def frame_data_packet(frame_id, timestamp, width, height, keyframe, pixels: bytes, sequence, packet_ts) -> bytes:
    body = struct.pack("<IQHHB", frame_id, timestamp, width, height, 1 if keyframe else 0) + pixels
    return packet(FRAME_DATA, sequence, packet_ts, body)


def detection_result_packet(frame_id, timestamp, dets_raw: bytes, count, sequence, packet_ts) -> bytes:
    body = struct.pack("<IQH", frame_id, timestamp, count) + dets_raw
    return packet(DETECTION_RESULT, sequence, packet_ts, body)


def check(raw: bytes, expected_type: int):
    """-> (error code or 0) as Packet::deserialize returns it: 103 INVALID_PACKET, 105 PROTOCOL_ERROR"""
    if len(raw) < HEADER:
        return 103
    magic, version, ptype, length, _seq, _ts, checksum = struct.unpack_from("<IBBHIQH", raw, 0)
    if magic != MAGIC or version != VERSION:
        return 105
    if HEADER + length != len(raw):
        return 103
    if ptype != expected_type:
        return 105
    tmp = bytearray(raw)
    tmp[20:22] = b"\0\0"
    return 0 if crc16(bytes(tmp[2:])) == checksum else 105


def frame_chunk_packets(frame_id, timestamp, width, height, keyframe, pixels: bytes, max_payload, sequence0, packet_ts):
    """EXTENSION, no reference counterpart (host/zly_wire.hpp FrameChunkPacket): a raw w*h*3 frame as ceil(n / max_payload) packets of
    type 8; body = the FrameDataPacket fields + chunk index u16, chunk count u16, byte offset u32, then the piece."""
    assert len(pixels) == width * height * 3 and 0 < max_payload <= 0xFFFF - 25
    count = (len(pixels) + max_payload - 1) // max_payload
    out = []
    for i in range(count):
        off = i * max_payload
        body = struct.pack("<IQHHBHHI", frame_id, timestamp, width, height, 1 if keyframe else 0, i, count, off) + pixels[off:off + max_payload]
        out.append(packet(FRAME_CHUNK, sequence0 + i, packet_ts, body))
    return out


def reassemble(packets):
    """pieces (any order, duplicates allowed) of ONE frame -> (frame_id, timestamp, width, height, keyframe, pixels) or None while incomplete"""
    meta, have, buf = None, {}, None
    for raw in packets:
        assert check(raw, FRAME_CHUNK) == 0
        frame_id, ts, w, h, key, idx, count, off = struct.unpack_from("<IQHHBHHI", raw, HEADER)
        if meta is None:
            meta, buf = (frame_id, ts, w, h, key, count), bytearray(w * h * 3)
        assert meta == (frame_id, ts, w, h, key, count)
        piece = raw[HEADER + 25:]
        buf[off:off + len(piece)] = piece
        have[idx] = len(piece)
    if meta is None or len(have) < meta[5] or sum(have.values()) != len(buf):
        return None
    return meta[0], meta[1], meta[2], meta[3], bool(meta[4]), bytes(buf)
