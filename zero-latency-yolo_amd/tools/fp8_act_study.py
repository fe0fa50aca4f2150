#!/usr/bin/env python3
"""BASELINE configs[4] (YOLOv8-s 640 x 640, fp8): what would e4m3 ACTIVATIONS cost in accuracy?  CPU study on the oracle (oracle/yolov8_ref.py: test
infrastructure; this tool only measures, nothing here ships).  The engine today stores fp8 weights and computes in bf16; a v_mfma_f32_16x16x128_f8f6f4 path
needs both operands in fp8, i.e. the producing layer's epilogue would round activations to e4m3 with a per-tensor power-of-two scale.  For the committed
synthetic fp8 weight file this prints the head error against the fp32 oracle (SURVEY 8c's bars: box <= 1.5 px, score <= 2e-2) for
  bf16            the engine's arithmetic today (bf16 activations, fp8 weights dequantised)
  fp8 3x3 dense   e4m3 inputs on the FLOP-dense 3x3 convs only (Cin >= 128: 62 % of the s model's MACs)
  fp8 3x3 all     e4m3 inputs on every 3x3 conv with Cin >= 32
  fp8 all         e4m3 inputs on every conv but the stem and the six final Detect convs
    python3 zero-latency-yolo_amd/tools/fp8_act_study.py [--frames 2] [--size 640] [--scale s]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("oracle", "tests", "zero-latency-yolo_amd/tools", "zero-latency-yolo_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import yolov8_ref          # noqa: E402
import zly_model as zm     # noqa: E402
from oracle_lib import Oracle   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--scale", default="s")
    a = ap.parse_args()
    path = os.path.join(ROOT, "zero-latency-yolo_amd", "_build", f"yolov8{a.scale}_synth_fp8.zlyw")
    if not os.path.exists(path):
        spec = zm.build_spec(a.scale)
        zm.write_zlyw(path, spec, zm.synth_weights(spec), fp8=True)
    frames = zm.synth_frames(a.frames, a.size, a.size, seed=20250328, rects=False)
    orc = Oracle()
    x = np.stack([orc.preprocess(f, a.size, a.size)[1] for f in frames])
    xt = torch.from_numpy(x)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) - 1))
    want = yolov8_ref.load(path, "fp32").forward(xt).numpy()
    ref = yolov8_ref.load(path, "bf16")
    spec = ref.spec
    macs = {n: c.cout * c.cin * c.k * c.k for n, c in spec.items()}          # per output pixel; relative weights only need the map sizes too
    final = {f"model.22.cv{b}.{l}.2" for b in (2, 3) for l in range(3)}
    modes = [("bf16 (today)", None),
             ("fp8 act, 3x3 convs with Cin >= 128", lambda n: spec[n].k == 3 and spec[n].cin >= 128),
             ("fp8 act, 3x3 convs with Cin >= 32", lambda n: spec[n].k == 3 and spec[n].cin >= 32),
             ("fp8 act, every conv but stem / final Detect", lambda n: n != "model.0" and n not in final)]
    print(f"YOLOv8-{a.scale} {a.size}x{a.size}, fp8 e4m3 weight file, {a.frames} synthetic frames; head [4+nc][N] against the fp32 oracle (bars: box max <= 1.5 px, score max <= 2e-2)")
    modes.append(("fp8 act, 3x3 Cin >= 128, per-CHANNEL scales", modes[1][1]))
    for label, pred in modes:
        ref.fp8_act = pred
        ref.fp8_per_channel = "per-CHANNEL" in label
        got = ref.forward(xt).numpy()
        db, ds = got[:, :4] - want[:, :4], got[:, 4:] - want[:, 4:]
        n_l = sum(1 for n in spec if pred and pred(n))
        print(f"  {label:46s} layers {n_l:2d}: box rms {np.sqrt(np.mean(db.astype(np.float64) ** 2)):.3f} px max {np.abs(db).max():.2f} px | "
              f"score rms {np.sqrt(np.mean(ds.astype(np.float64) ** 2)):.5f} max {np.abs(ds).max():.4f}")


if __name__ == "__main__":
    main()
