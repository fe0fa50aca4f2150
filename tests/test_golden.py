"""Committed golden fixtures (tests/golden/zly_golden_64.npz, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them.  GPU (-m gpu): the HIP engine, built for a 64x64 model input (feature
maps 8x8 / 4x4 / 2x2: every conv tile is larger than the map, the SPPF window larger than the map), matches
them without the oracle being run on the box."""
import hashlib
import os

import numpy as np
import pytest
import torch

import zly_model as zm
from oracle_lib import det_fields_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "zly_golden_64.npz"))
CONF, IOU = float(G["conf"]), float(G["iou"])


def _frames():
    return [G["frames_64"][0], G["frames_64"][1], G["frame_96x80"]]


def test_weights_file_is_the_one_the_fixtures_were_made_with(weights_path):
    sha = hashlib.sha256(open(weights_path, "rb").read()).hexdigest()
    assert sha == bytes(G["weights_sha256"]).decode()


def test_oracle_reproduces_golden(oracle, ref_fp32):
    frames = _frames()
    pre = np.stack([oracle.preprocess(f, 64, 64)[1] for f in frames])
    assert np.array_equal(pre, G["pre"])
    head = ref_fp32.forward(torch.from_numpy(pre)).numpy()
    assert np.abs(head - G["head"]).max() < 1e-3           # torch build / thread-count differences only
    for i, f in enumerate(frames):
        d = oracle.postprocess(G["head"][i], f.shape[1], f.shape[0], CONF, IOU)
        assert det_fields_equal(d, G[f"dets_{i}"]) and len(d) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_engine_matches_golden(weights_path, dtype):
    import zly
    eng = zly.Engine(weights_path, model_w=64, model_h=64, conf_thr=CONF, iou_thr=IOU, max_batch=3, max_dets=256,
                     dtype=zly.DTYPE_FP32 if dtype == "fp32" else zly.DTYPE_BF16, warmup_runs=1)
    assert eng.N == 84
    frames = _frames()
    for i, f in enumerate(frames):
        assert np.array_equal(eng.preprocess(f), G["pre"][i])                    # bit-exact
        got, n_kept, _ = eng.postprocess(G["head"][i], f.shape[1], f.shape[0], CONF, IOU)
        assert n_kept == len(G[f"dets_{i}"]) and det_fields_equal(got, G[f"dets_{i}"])   # bit-exact
    head = eng.forward(G["pre"])
    d = np.abs(head - G["head"])
    if dtype == "fp32":
        assert d[:, :4].max() <= 1e-3 and d[:, 4:].max() <= 1e-4
        res = eng.detect_batch(frames, cap=256)
        for i, (dets, n) in enumerate(res):                                      # whole path, three frames, two sizes
            want = G[f"dets_{i}"]
            assert n == len(want) and np.array_equal(dets["class_id"], want["class_id"])
            assert np.abs(dets["confidence"] - want["confidence"]).max() <= 1e-4
            for k in ("x", "y", "w", "h"):
                assert np.abs(dets[k] - want[k]).max() <= 2e-3
    else:
        assert d[:, :4].max() <= 1.5 and d[:, 4:].max() <= 2e-2                    # bf16 tolerance
        res = eng.detect_batch(frames, cap=256)
        assert all(n > 0 for _, n in res)
    eng.close()
