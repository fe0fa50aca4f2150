#!/usr/bin/env python3
"""Generates tests/golden/zly_golden_64.npz with the CPU oracle (oracle/zly_oracle.c + oracle/yolov8_ref.py).

The reference ships no golden vectors and cannot be run, so these fixtures are produced
by the oracle that the KATs pin to the reference source; they freeze its outputs on small seeded inputs so that
(a) the oracle itself cannot drift unnoticed (tests/test_golden.py, CPU) and (b) the HIP engine can be checked on
the GPU box against committed data without re-running the oracle (tests/test_golden.py, -m gpu).

Contents (model input 64x64 -> 84 anchors, the seeded synthetic yolov8n weights):
  frames_64   u8 [2][64][64][3]      frame_96x80  u8 [80][96][3]  (exercises the stretch-resize path)
  pre         fp32 [3][3][64][64]    preProcess of the three frames
  head        fp32 [3][84][84]       fp32 oracle forward
  dets_*      structured zly_det arrays: postProcess + NMS at conf 0.17 / IoU 0.45, request dims as given
  weights_sha256                      of the generated ZLYW file
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("zero-latency-yolo_amd/tools", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import zly_model as zm          # noqa: E402
import yolov8_ref               # noqa: E402
from oracle_lib import Oracle   # noqa: E402

CONF, IOU = 0.17, 0.45      # scores on 64x64 noise frames are 0.09 .. 0.33 (median 0.16): 0.17 keeps ~1/3 of the anchors


def inputs():
    frames_64 = zm.synth_frames(2, 64, 64, seed=1234, rects=False)
    frame_96x80 = zm.synth_frames(1, 96, 80, seed=4321, rects=False)[0]
    return frames_64, frame_96x80


def main():
    wpath = os.path.join(ROOT, "zero-latency-yolo_amd", "_build", "yolov8n_synth.zlyw")
    spec = zm.build_spec("n")
    if not os.path.exists(wpath):
        os.makedirs(os.path.dirname(wpath), exist_ok=True)
        zm.write_zlyw(wpath, spec, zm.synth_weights(spec))
    sha = hashlib.sha256(open(wpath, "rb").read()).hexdigest()
    orc = Oracle()
    ref = yolov8_ref.load(wpath, "fp32")
    frames_64, frame_96x80 = inputs()
    frames = [frames_64[0], frames_64[1], frame_96x80]
    pre = np.stack([orc.preprocess(f, 64, 64)[1] for f in frames])
    head = ref.forward(torch.from_numpy(pre)).numpy()
    out = dict(frames_64=frames_64, frame_96x80=frame_96x80, pre=pre, head=head.astype(np.float32),
               weights_sha256=np.frombuffer(sha.encode(), dtype=np.uint8), conf=np.float32(CONF), iou=np.float32(IOU))
    for i, f in enumerate(frames):
        d = orc.postprocess(head[i], f.shape[1], f.shape[0], CONF, IOU)
        out[f"dets_{i}"] = d
        print(f"frame {i}: {f.shape[1]}x{f.shape[0]} -> {len(orc.decode(head[i], f.shape[1], f.shape[0], CONF))} candidates, {len(d)} detections")
    path = os.path.join(ROOT, "tests", "golden", "zly_golden_64.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
