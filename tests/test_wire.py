"""Wire format either side of the detect path: host/zly_wire.hpp against the Python
restatement oracle/wire_ref.py, byte for byte, plus known answers derived from reference src/common/protocol.h."""
import os
import struct
import subprocess

import numpy as np

import wire_ref
from oracle_lib import DET_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "zero-latency-yolo_amd", "_build", "test_wire")


def test_crc_is_ccitt_false_over_bytes_2_onwards():
    assert wire_ref.crc16(b"123456789") == 0x29B1                 # poly 0x1021, init 0xFFFF, no reflection (protocol.h:76-89)
    p = wire_ref.packet(0, 7, 1000, b"\x01\x02\x03\x04")
    assert len(p) == 22 + 4 and struct.unpack_from("<I", p, 0)[0] == 0x59544C5A and p[4] == 1 and p[5] == 0
    assert struct.unpack_from("<H", p, 6)[0] == 4 and struct.unpack_from("<I", p, 8)[0] == 7
    z = bytearray(p); z[20:22] = b"\0\0"
    assert struct.unpack_from("<H", p, 20)[0] == wire_ref.crc16(bytes(z[2:]))   # skips magic[0:2] (protocol.h:182-185)
    assert wire_ref.check(p, 0) == 0 and wire_ref.check(p, 3) == 105 and wire_ref.check(p[:-1], 0) == 103


def test_cpp_wire_matches_oracle(tmp_path):
    if not os.path.exists(BIN):
        subprocess.run(["make", "-C", ROOT, "host"], check=True, stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(5)
    w, h = 40, 30
    pixels = rng.integers(0, 256, w * h * 3, dtype=np.uint8).tobytes()
    pkt = wire_ref.frame_data_packet(frame_id=9001, timestamp=1700000000123, width=w, height=h, keyframe=True,
                                     pixels=pixels, sequence=31337, packet_ts=1700000000999)
    (tmp_path / "frame.pkt").write_bytes(pkt)
    r = subprocess.run([BIN, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    rep = dict(line.split("=", 1) for line in (tmp_path / "report.txt").read_text().splitlines())
    assert rep["parse"] == "0" and rep["frame_id"] == "9001" and rep["timestamp"] == "1700000000123"
    assert (rep["width"], rep["height"], rep["keyframe"], rep["bytes"]) == ("40", "30", "1", str(w * h * 3))
    assert rep["sequence"] == "31337" and rep["packet_ts"] == "1700000000999"
    assert (tmp_path / "frame_roundtrip.pkt").read_bytes() == pkt                      # serialise(parse(x)) == x
    # handleFrameData (network_server.cpp:184-207)
    assert rep["request"] == "0" and rep["req_client"] == "42" and rep["req_frame"] == "9001" and rep["req_keyframe"] == "1"
    assert rep["request_short"] == "203" and rep["request_empty"] == "203"             # INVALID_INPUT
    # Packet::deserialize's checks, same codes as the oracle
    bad = bytearray(pkt); bad[30] ^= 1
    assert rep["bad_crc"] == str(wire_ref.check(bytes(bad), 3)) == "105"
    assert rep["bad_magic"] == "105" and rep["bad_length"] == "103" and rep["bad_type"] == "105" and rep["too_small"] == "103"
    # DetectionResultPacket: 3 detections, raw 40-byte records (protocol.h:541-567)
    dets = np.zeros(3, dtype=DET_DTYPE)
    for i in range(3):
        dets[i] = (np.float32(0.1) * np.float32(i + 1), np.float32(0.2), np.float32(0.3), np.float32(0.4),
                   np.float32(0.5 + 0.125 * i), i * 7, 100 + i, 0, 999000 + i)
    want = wire_ref.detection_result_packet(77, 1234567890123, dets.tobytes(), 3, sequence=5, packet_ts=424242)
    got = (tmp_path / "dets.pkt").read_bytes()
    assert got == want and len(got) == 22 + 14 + 3 * 40
    assert rep["dets_serialize"] == "0" and rep["dets_parse"] == "0" and rep["dets_count"] == "3"
    # SHA-256 of the plugin's model-file watch (host/zly_sha256.hpp; the reference uses OpenSSL, onnx_engine.cpp:1087-1124)
    import hashlib
    assert rep["sha_abc"] == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"
    assert rep["sha_empty"] == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"
    assert rep["sha_56"] == "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"
    assert rep["sha_frame_pkt"] == hashlib.sha256(pkt).hexdigest() and rep["sha_missing"] == ""
    assert rep["dets_too_big"] == "104"                                                # PACKET_TOO_LARGE instead of a truncated length


def test_chunked_raw_frame_reassembly_matches_oracle(tmp_path):
    """The frame that fits the wire (SURVEY 8f rank 1 follow-up): a raw 416x416 frame (519 168 bytes) cannot travel in one
    FrameDataPacket (16-bit length, protocol.h:42).  FrameChunkPacket (extension, type 8) cuts it into pieces; host/zly_wire.hpp's
    FrameAssembler must rebuild the exact frame from the pieces in any order, with duplicates, and cut a frame into the very bytes
    the Python restatement produces."""
    if not os.path.exists(BIN):
        subprocess.run(["make", "-C", ROOT, "host"], check=True, stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(11)
    w, h = 416, 416
    pixels = rng.integers(0, 256, w * h * 3, dtype=np.uint8).tobytes()
    small = wire_ref.frame_data_packet(1, 1, 8, 8, False, bytes(192), 1, 1)
    (tmp_path / "frame.pkt").write_bytes(small)                                             # the driver's other sections need it
    pk = wire_ref.frame_chunk_packets(frame_id=4242, timestamp=1700000000777, width=w, height=h, keyframe=True, pixels=pixels,
                                      max_payload=60000, sequence0=700, packet_ts=555555)
    assert len(pk) == 9 and all(len(p) <= 22 + 25 + 60000 for p in pk) and wire_ref.check(pk[0], wire_ref.FRAME_CHUNK) == 0
    assert wire_ref.reassemble(pk[:-1]) is None and wire_ref.reassemble(pk) == (4242, 1700000000777, w, h, True, pixels)
    order = [5, 0, 8, 0, 3, 1, 7, 2, 6, 4]                                                  # shuffled arrival, piece 0 twice
    arrive = [pk[i] for i in order]
    (tmp_path / "chunks.bin").write_bytes(struct.pack("<I", len(arrive)) + b"".join(struct.pack("<I", len(p)) + p for p in arrive))
    r = subprocess.run([BIN, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    rep = dict(line.split("=", 1) for line in (tmp_path / "report.txt").read_text().splitlines())
    assert rep["chunks_completed"] == "1" and rep["chunks_errors"] == "0" and rep["chunks_complete_at"] == str(len(order) - 1) and rep["chunks_pending"] == "0"
    assert (rep["chunks_frame_id"], rep["chunks_ts"], rep["chunks_w"], rep["chunks_h"], rep["chunks_key"]) == ("4242", "1700000000777", "416", "416", "1")
    assert (tmp_path / "chunks_frame.bin").read_bytes() == pixels                           # byte-exact reassembly
    out = (tmp_path / "chunks_out.bin").read_bytes()
    o, got = 0, []
    while o < len(out):
        nb = struct.unpack_from("<I", out, o)[0]; o += 4
        got.append(out[o:o + nb]); o += nb
    assert rep["chunks_serialize"] == "0" and got == pk                                     # the C++ cutter == the oracle, byte for byte
    assert rep["chunk_past_end"] == "103" and rep["chunk_bad_index"] == "103" and rep["chunk_contradicts"] == "103"    # INVALID_PACKET
    assert rep["chunk_evicted"] == "1" and rep["chunk_pending_after"] == "2"
    # non-uniform tilings are refused by the parser; a datagram that claims a 2.7 GB frame allocates nothing
    assert rep["chunk_overlap"] == "103" and rep["chunk_bad_count"] == "103"
    assert rep["chunk_huge_parse"] == "0" and rep["chunk_huge_add"] == "104" and rep["chunk_huge_pending"] == "0" and rep["chunk_limit_exact"] == "1"


def test_wire_oracle_matches_committed_vectors():
    """tests/golden/wire_golden.json (made by tests/golden/make_wire_golden.py) pins the Python restatement itself."""
    import importlib.util
    import json
    g = os.path.join(ROOT, "tests", "golden")
    spec = importlib.util.spec_from_file_location("make_wire_golden", os.path.join(g, "make_wire_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(g, "wire_golden.json")))
    assert mod.vectors() == want
    assert want["crc16_123456789"] == 0x29B1 and len(bytes.fromhex(want["empty_heartbeat"])) == 22
