"""CPU checks of the set-level detection comparison (tests/parity_sets.py) that the GPU parity tests
use: the bf16-rounding oracle stands in for the engine (same rounding points, on the CPU), so the comparison must
pass on it with most detections compared exactly, and must fail on an injected few-pixel box shift, a dropped and an
invented detection."""
import numpy as np
import torch

import zly_model as zm
from parity_sets import compare_detection_sets


def _heads(oracle, ref_fp32, ref_bf16, frames):
    x = np.stack([oracle.preprocess(f, 416, 416)[1] for f in frames])
    t = torch.from_numpy(x)
    return ref_fp32.forward(t).numpy(), ref_bf16.forward(t).numpy()


def test_bf16_oracle_detections_match_fp32_oracle_detections(oracle, ref_fp32, ref_bf16):
    frames = list(zm.synth_frames(15, 416, 416, seed=5, rects=False)) + [zm.synth_frames(1, 800, 600, seed=3, rects=False)[0]]
    h32, h16 = _heads(oracle, ref_fp32, ref_bf16, frames)
    # the noise floor of bf16 itself on the synthetic model is inside SURVEY 8c's tolerance (box <= 1.5 px, score <= 2e-2)
    assert np.abs(h32[:, :4] - h16[:, :4]).max() <= 1.5 and np.abs(h32[:, 4:] - h16[:, 4:]).max() <= 2e-2
    compared = skipped = 0
    for f, a, b in zip(frames, h32, h16):
        got = oracle.postprocess(b, f.shape[1], f.shape[0])
        c, s, errors = compare_detection_sets(oracle, a, got, f.shape[1], f.shape[0])             # SURVEY's fixed bands: a minority is compared
        assert not errors, errors
        c2, s2, errors = compare_detection_sets(oracle, a, got, f.shape[1], f.shape[0], got_head=b)  # the decisions that really differ: most are
        assert not errors, errors
        assert c + s == c2 + s2 and c2 >= c
        compared += c2; skipped += s2
    assert compared >= 100 and compared >= 0.7 * (compared + skipped), (compared, skipped)


def test_comparison_catches_shifted_dropped_and_invented_detections(oracle, ref_fp32):
    f = zm.synth_frames(1, 416, 416, seed=5, rects=False)[0]
    head = ref_fp32.forward(torch.from_numpy(oracle.preprocess(f, 416, 416)[1][None])).numpy()[0]
    want = oracle.postprocess(head, 416, 416)
    c, s, errors = compare_detection_sets(oracle, head, want, 416, 416)
    assert not errors and c + s == len(want) and c > 0
    # find a detection that sits in an unambiguous component: shifting it must be seen
    for k in range(len(want)):
        bad = want.copy()
        bad["x"][k] += 6.0 / 416.0                      # 6 px of a ~64 px box: IoU drops below 0.9
        _, _, e1 = compare_detection_sets(oracle, head, bad, 416, 416)
        if e1:
            break
    assert e1, "a 6 px shift of a detection went unnoticed"
    dropped = 0
    for k in range(len(want)):
        _, _, e2 = compare_detection_sets(oracle, head, np.delete(want, k), 416, 416)
        dropped += bool(e2)
    assert dropped >= c                                 # every exactly-compared detection is missed when absent
    # ... and the same with the engine-head form of the comparison (here: the oracle's own head, so nothing differs and nothing is skipped)
    c4, s4, e4 = compare_detection_sets(oracle, head, want, 416, 416, got_head=head)
    assert not e4 and c4 + s4 == len(want) and c4 >= c
    _, _, e5 = compare_detection_sets(oracle, head, np.delete(want, 0), 416, 416, got_head=head)
    assert e5 or s4 > 0
    ghost = want[:1].copy()
    ghost["x"] = 0.5; ghost["y"] = 0.5; ghost["w"] = 0.9; ghost["h"] = 0.9
    _, _, e3 = compare_detection_sets(oracle, head, np.concatenate([want, ghost]), 416, 416)
    assert e3
