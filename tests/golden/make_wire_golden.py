"""Writes tests/golden/wire_golden.json: packets produced by the wire-format oracle (oracle/wire_ref.py, a restatement of
reference src/common/protocol.h) for fixed inputs, as hex -- pins the oracle itself against silent edits; the C++ side
(host/zly_wire.hpp) is compared with the oracle byte for byte in tests/test_wire.py.  Run from the repo root."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import wire_ref                      # noqa: E402
from oracle_lib import DET_DTYPE     # noqa: E402


def vectors():
    rng = np.random.default_rng(2025)
    px = rng.integers(0, 256, 4 * 3 * 3, dtype=np.uint8).tobytes()
    dets = np.zeros(2, dtype=DET_DTYPE)
    dets[0] = (0.25, 0.5, 0.125, 0.0625, 0.75, 2, 11, 0, 1700000000001)
    dets[1] = (0.5, 0.5, 0.5, 0.5, 0.5, 0, 12, 0, 1700000000002)
    return {
        "crc16_123456789": wire_ref.crc16(b"123456789"),
        "empty_heartbeat": wire_ref.packet(0, 1, 2, b"").hex(),
        "frame_4x3": wire_ref.frame_data_packet(7, 1700000000000, 4, 3, True, px, 42, 1700000000123).hex(),
        "result_2_dets": wire_ref.detection_result_packet(7, 1700000000000, dets.tobytes(), 2, 43, 1700000000456).hex(),
        "result_0_dets": wire_ref.detection_result_packet(8, 5, b"", 0, 44, 6).hex(),
    }


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "wire_golden.json")
    json.dump(vectors(), open(out, "w"), indent=1, sort_keys=True)
    print(out)
