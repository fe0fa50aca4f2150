"""Post-step after the engine callback: host/zly_game_step.hpp against the Python
restatement oracle/game_step_ref.py, record for record, plus hand-derived known answers from
reference src/game/games/cs16/cs16_game_adapter.cpp:36-69,243-262."""
import os
import struct
import subprocess

import numpy as np

import game_step_ref
from oracle_lib import DET_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "zero-latency-yolo_amd", "_build", "test_game_step")


def det(x, y, w, h, conf, cls, track, ts):
    d = np.zeros(1, dtype=DET_DTYPE)
    d[0] = (x, y, w, h, conf, cls, track, 0, ts)
    return d


def test_known_answers():
    ref = game_step_ref.Cs16StepRef()
    raw = np.concatenate([det(.5, .5, .2, .4, .9, 0, 0, 1000), det(.5, .3, .1, .5, .8, 2, 0, 1000), det(.1, .1, .1, .1, .7, 2, 77, 1000)])
    assert ref.process(1, 1, 1000, raw)[0] == 3                          # NOT_INITIALIZED before initialize()
    ref.initialized = True
    assert ref.process(1, 2, 1000, raw)[0] == 2                          # CSGO id on the CS 1.6 adapter
    code, out = ref.process(1, 1, 1000, raw)
    assert code == 0 and list(out["track_id"]) == [1, 2, 77]             # ids from 1, a preset id is kept
    assert out["h"][0] == np.float32(.4) and out["h"][1] == np.float32(.5) * np.float32(.7) and out["h"][2] == np.float32(.1) * np.float32(.7)
    assert sorted(ref.clients[1]) == [1, 2, 77]
    code, out = ref.process(1, 1, 1101, det(.5, .5, .2, .4, .9, 0, 1, 1101))      # 101 ms later: 2 and 77 expire, 1 was refreshed
    assert sorted(ref.clients[1]) == [1]
    code, out = ref.process(2, 1, 50, det(.5, .5, .2, .4, .9, 0, 0, 60))          # stamped AFTER the frame: unsigned wrap -> expires at once
    assert list(out["track_id"]) == [3] and ref.clients[2] == {}                  # the counter is shared by all clients


def test_cpp_step_matches_oracle(tmp_path):
    if not os.path.exists(BIN):
        subprocess.run(["make", "-C", ROOT, "host"], check=True, stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(11)
    ref = game_step_ref.Cs16StepRef()
    frames, blob = [], [struct.pack("<I", 60)]
    t = 5000
    for i in range(60):
        init_now = 1 if i == 2 else 0
        client = int(rng.integers(1, 4))
        game = 1 if rng.random() > 0.1 else 2
        t += int(rng.integers(0, 90))
        n = int(rng.integers(0, 6))
        d = np.zeros(n, dtype=DET_DTYPE)
        for k in range(n):
            d[k] = (rng.random(dtype=np.float32), rng.random(dtype=np.float32), rng.random(dtype=np.float32), rng.random(dtype=np.float32),
                    rng.random(dtype=np.float32), int(rng.integers(0, 4)), int(rng.integers(1, 8)) if rng.random() < 0.3 else 0, 0,
                    t - int(rng.integers(-20, 150)))
        frames.append((init_now, client, game, t, d))
        blob.append(struct.pack("<BIBIQH", init_now, client, game, 100 + i, t, n) + d.tobytes())
    (tmp_path / "scenario.bin").write_bytes(b"".join(blob))
    r = subprocess.run([BIN, str(tmp_path / "scenario.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    got = (tmp_path / "out.bin").read_bytes()
    o = 0
    n_ok = 0
    for init_now, client, game, ts, d in frames:
        if init_now:
            ref.initialized = True
        code, out = ref.process(client, game, ts, d)
        gcode, gcount = struct.unpack_from("<iH", got, o); o += 6
        assert gcode == code
        if code == 0:
            n_ok += 1
            assert gcount == len(out) and got[o:o + 40 * gcount] == out.tobytes()        # byte-identical records
            o += 40 * gcount
        tracked, next_id = struct.unpack_from("<II", got, o); o += 8
        assert tracked == len(ref.clients.get(client, {})) and next_id == ref.next_track_id
    assert o == len(got) and n_ok > 40
