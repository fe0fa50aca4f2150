"""CPU ORACLE (test infrastructure) for the post-step after the engine callback: a plain-Python restatement of
CS16GameAdapter::processDetections / processCS16Detections (reference src/game/games/cs16/cs16_game_adapter.cpp:36-69,
243-262) with the tracked-object table of src/game/base/game_adapter_base.h:64-115.  Detections are numpy records of the
40-byte layout; float work is one fp32 multiply.  Pinned by hand-derived known answers in tests/test_game_step.py
(the reference ships no test for it)."""
import numpy as np

CS_1_6, CLASS_HEAD = 1, 2
MASK64 = (1 << 64) - 1


class Cs16StepRef:
    def __init__(self, head_size_factor=0.7):
        self.f = np.float32(head_size_factor)
        self.next_track_id = 1
        self.clients = {}
        self.initialized = False

    def process(self, client_id, game_id, timestamp, dets):
        """-> (error code, processed records)"""
        if not self.initialized:
            return 3, None
        if game_id != CS_1_6:
            return 2, None
        out = dets.copy()
        for d in out:
            if d["track_id"] == 0:
                d["track_id"] = self.next_track_id
                self.next_track_id = (self.next_track_id + 1) & 0xFFFFFFFF
            if d["class_id"] == CLASS_HEAD:
                d["h"] = np.float32(d["h"]) * self.f
        tracked = self.clients.setdefault(client_id, {})
        for d in out:
            tracked[int(d["track_id"])] = d.copy()
        for tid in [t for t, d in tracked.items() if ((int(timestamp) - int(d["timestamp"])) & MASK64) > 100]:
            del tracked[tid]
        return 0, out
