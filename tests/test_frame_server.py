"""Bytes in, bytes out around the plugin (SURVEY 8f rank 1 + 2 end to end): FrameDataPacket datagrams -> host/zly_frame_server.hpp
(NetworkServer::handleFrameData / onInferenceResult, reference src/network/network_server.cpp:184-283) -> HipInferenceEngine ->
game-adapter step -> DetectionResultPacket datagrams.  Checked against the Python restatements (oracle/wire_ref.py,
oracle/game_step_ref.py) and the ctypes engine."""
import os
import struct
import subprocess

import numpy as np
import pytest

import game_step_ref
import wire_ref
import zly_model as zm
from oracle_lib import DET_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "zero-latency-yolo_amd", "_build", "test_frame_server")


@pytest.mark.gpu
def test_packets_in_packets_out(tmp_path, weights_path):
    import zly
    if not os.path.exists(BIN):
        subprocess.run(["make", "-C", ROOT, "host"], check=True, stdout=subprocess.DEVNULL)
    w, h = 128, 96                                                # 36 864 raw bytes: fits the 16-bit packet length
    frames = zm.synth_frames(6, w, h, seed=77, rects=False)
    blob, sent = [], []
    for i, f in enumerate(frames):
        pkt = wire_ref.frame_data_packet(frame_id=500 + i, timestamp=9_000_000 + 40 * i, width=w, height=h, keyframe=(i == 0),
                                         pixels=f.tobytes(), sequence=i, packet_ts=123456 + i)
        client = 7 + i % 2
        if i == 2:
            pkt = pkt[:40] + bytes([pkt[40] ^ 0xFF]) + pkt[41:]   # corrupted payload: checksum mismatch, dropped by the server
        if i == 4:
            pkt = wire_ref.frame_data_packet(frame_id=504, timestamp=1, width=w, height=h, keyframe=False, pixels=f.tobytes()[:-3],
                                             sequence=i, packet_ts=1)                      # short frame: INVALID_INPUT in handleFrameData
        blob.append(struct.pack("<II", client, len(pkt)) + pkt)
        sent.append((client, 500 + i, 9_000_000 + 40 * i, i not in (2, 4)))
    # a REAL 416x416 frame (519 168 bytes: does not fit one FrameDataPacket) as 9 FrameChunkPackets, arriving shuffled and with a duplicate
    big = zm.synth_frames(1, 416, 416, seed=78, rects=False)[0]
    pieces = wire_ref.frame_chunk_packets(frame_id=600, timestamp=9_100_000, width=416, height=416, keyframe=False, pixels=big.tobytes(),
                                          max_payload=60000, sequence0=50, packet_ts=777)
    for j in (4, 0, 8, 2, 2, 6, 1, 7, 3, 5):
        blob.append(struct.pack("<II", 7, len(pieces[j])) + pieces[j])
    sent.append((7, 600, 9_100_000, True))
    frames = list(frames) + [big]
    (tmp_path / "in.bin").write_bytes(struct.pack("<I", len(blob)) + b"".join(blob))
    env = dict(os.environ, ZLY_MAX_BATCH="1", ZLY_MODEL_WATCH_MS="0")
    r = subprocess.run([BIN, weights_path, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    raw = (tmp_path / "out.bin").read_bytes()
    m = struct.unpack_from("<I", raw, 0)[0]
    o = 4
    eng = zly.Engine(weights_path, model_w=w, model_h=h, conf_thr=0.02, max_batch=1, max_dets=256, warmup_runs=1)
    step = game_step_ref.Cs16StepRef()
    step.initialized = True
    good = [(c, fid, ts, i) for i, (c, fid, ts, ok) in enumerate(sent) if ok]
    assert m == len(good)
    total = 0
    for k, (client, fid, ts, i) in enumerate(good):
        c, nb = struct.unpack_from("<II", raw, o); o += 8
        pkt = raw[o:o + nb]; o += nb
        assert c == client and wire_ref.check(pkt, wire_ref.DETECTION_RESULT) == 0
        assert struct.unpack_from("<I", pkt, 8)[0] == k                                  # server-side sequence numbers
        frame_id, stamp, count = struct.unpack_from("<IQH", pkt, 22)
        assert (frame_id, stamp) == (fid, ts) and len(pkt) == 22 + 14 + 40 * count
        got = np.frombuffer(pkt[36:], dtype=DET_DTYPE)
        dets, n = eng.detect(frames[i], cap=256)
        code, want = step.process(client, 1, ts, dets[:n].copy())
        assert code == 0 and count == n
        for name in DET_DTYPE.names:                              # track ids, HEAD scaling, raw records
            if name != "timestamp":                              # wall clock of the call that produced it (onnx_engine.cpp:812)
                assert np.array_equal(got[name], want[name]), (k, name)
        total += n
    assert struct.unpack_from("<Q", raw, o)[0] == 2 and total > 0                        # the two bad datagrams were counted, not served
    assert struct.unpack_from("<Q", raw, o + 8)[0] == 1                                  # the chunked 416x416 frame was reassembled and served
    eng.close()
