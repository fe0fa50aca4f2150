"""CPU ORACLE for the YOLOv8 forward pass (the part the reference delegates to ONNX Runtime).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The HIP engine never does.

PARITY UNPINNED for this part.  The reference contains no neural-network arithmetic: it
calls `Ort::Session::Run` (reference src/inference/onnx_engine.cpp:578-585) from un-vendored
ONNX Runtime v1.8.1 (reference start.sh:74) on a `yolov8n.onnx` exported at install time by
an unpinned `ultralytics` git HEAD (start.sh:97,109,122-125).  Neither library, nor any
.onnx/.pt file, nor any test/fixture at that boundary exists offline.  This file therefore
restates the PUBLISHED YOLOv8 detect architecture (yolov8.yaml + the Conv/C2f/Bottleneck/SPPF/
Detect/DFL module definitions) in plain PyTorch-CPU fp32, anchored on the reference's own
call sites for the tensor contract:
    input  "images"  fp32 [1,3,H,W] RGB in [0,1]              onnx_engine.cpp:49,560-569
    output "output0" fp32 [1,4+nc,N]; rows 0-3 = cx,cy,w,h in model-input pixels,
           rows 4.. = sigmoid class scores, channel-major     onnx_engine.cpp:50,767-796
The only external cross-checks available offline are ultralytics' published model sizes
(3.2 M params / 8.7 GFLOPs for n at 640, 11.2 M / 28.6 for s), which the layer table
reproduces (tests/test_model_spec.py).

Two numeric modes:
  mode="fp32"  plain fp32 everywhere: the target the north-star tolerance is stated against.
  mode="bf16"  emulates the HIP bf16 path's rounding points (weights and every stored
               activation rounded to bf16, fp32 accumulate, the six final Detect convs kept
               in fp32) so the bf16 engine can be checked to a few bf16 ulps, which separates
               indexing bugs from rounding.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class YoloV8Ref:
    def __init__(self, meta: dict, weights: Dict[str, Tuple[np.ndarray, np.ndarray]], mode: str = "fp32"):
        assert mode in ("fp32", "bf16")
        self.mode = mode
        self.nc = int(meta["nc"])
        self.reg_max = int(meta["reg_max"])
        self.ch = tuple(meta["ch"])
        self.n_c2f = tuple(meta["n_c2f"])
        self.spec = {c.name: c for c in meta["convs"]}
        self.w = {}
        for name, (w, b) in weights.items():
            wt = torch.from_numpy(np.array(w, dtype=np.float32))
            if mode == "bf16":
                wt = _bf16_round(wt)
            self.w[name] = (wt, torch.from_numpy(np.array(b, dtype=np.float32)))
        self.taps: Dict[str, torch.Tensor] = {}
        # calibration mode (oracle/calibrate_synth.py only): rescale each conv's weights on the fly so
        # that its pre-activation std hits a target; the factors are recorded in self.calib_scale.
        self.calib_target = None
        self.calib_scale: Dict[str, float] = {}
        # fp8 study (tools/fp8_act_study.py only): convs whose INPUT activations are rounded to OCP e4m3 with one power-of-two scale per tensor before
        # the conv -- what a v_mfma_f32_16x16x128_f8f6f4 path with activations quantised in the producer's epilogue would compute (BASELINE configs[4])
        self.fp8_act = None                     # predicate: conv name -> bool

    # -- building blocks ---------------------------------------------------------------------
    def _q(self, t):
        return _bf16_round(t) if self.mode == "bf16" else t

    def conv(self, name, x, residual=None, keep_fp32=False):
        c = self.spec[name]
        w, b = self.w[name]
        if self.fp8_act is not None and self.fp8_act(name):
            if getattr(self, "fp8_per_channel", False):                              # one power-of-two scale per input channel (what a per-channel epilogue scale could give)
                amax = x.abs().amax(dim=(0, 2, 3), keepdim=True).clamp_min(1e-30)
                sc = torch.exp2(torch.floor(torch.log2(448.0 / amax)))
            else:
                amax = float(x.abs().max())
                sc = 2.0 ** np.floor(np.log2(448.0 / amax)) if amax > 0 else 1.0      # e4m3fn: largest finite value 448
            x = (x * sc).to(torch.float8_e4m3fn).to(torch.float32) / sc
        y = F.conv2d(x, w, b, stride=c.stride, padding=c.k // 2)
        if self.calib_target is not None:
            tgt = self.calib_target(name)
            sc = float(tgt / (y - b.reshape(1, -1, 1, 1)).std())
            w.mul_(sc)
            self.calib_scale[name] = sc
            y = F.conv2d(x, w, b, stride=c.stride, padding=c.k // 2)
        if c.act:
            y = y * torch.sigmoid(y)            # SiLU
        if residual is not None:
            y = y + residual                     # Bottleneck shortcut: x + cv2(cv1(x))
        y = y if keep_fp32 else self._q(y)
        self.taps[name] = y
        return y

    def c2f(self, prefix, x, n, shortcut):
        y = self.conv(f"{prefix}.cv1", x)
        c = y.shape[1] // 2
        ys = [y[:, :c], y[:, c:]]
        for i in range(n):
            t = self.conv(f"{prefix}.m.{i}.cv1", ys[-1])
            ys.append(self.conv(f"{prefix}.m.{i}.cv2", t, residual=ys[-1] if shortcut else None))
        return self.conv(f"{prefix}.cv2", torch.cat(ys, 1))

    def sppf(self, x):
        y = self.conv("model.9.cv1", x)
        p1 = F.max_pool2d(y, 5, 1, 2)
        p2 = F.max_pool2d(p1, 5, 1, 2)
        p3 = F.max_pool2d(p2, 5, 1, 2)
        return self.conv("model.9.cv2", torch.cat([y, p1, p2, p3], 1))

    # -- forward -----------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, images: torch.Tensor) -> torch.Tensor:
        """images: fp32 [B,3,H,W] in [0,1]  ->  fp32 [B, 4+nc, N]."""
        n = self.n_c2f
        x = self._q(images)
        x = self.conv("model.0", x)
        x = self.conv("model.1", x)
        x = self.c2f("model.2", x, n[0], True)
        x = self.conv("model.3", x)
        p3 = self.c2f("model.4", x, n[1], True)
        x = self.conv("model.5", p3)
        p4 = self.c2f("model.6", x, n[2], True)
        x = self.conv("model.7", p4)
        x = self.c2f("model.8", x, n[3], True)
        p5 = self.sppf(x)
        up = F.interpolate(p5, scale_factor=2, mode="nearest")
        t12 = self.c2f("model.12", torch.cat([up, p4], 1), n[4], False)
        up = F.interpolate(t12, scale_factor=2, mode="nearest")
        t15 = self.c2f("model.15", torch.cat([up, p3], 1), n[5], False)
        x = self.conv("model.16", t15)
        t18 = self.c2f("model.18", torch.cat([x, t12], 1), n[6], False)
        x = self.conv("model.19", t18)
        t21 = self.c2f("model.21", torch.cat([x, p5], 1), n[7], False)
        return self.detect([t15, t18, t21])

    def detect(self, feats):
        B = feats[0].shape[0]
        boxes, clss, anchors, strides = [], [], [], []
        for lvl, f in enumerate(feats):
            b = self.conv(f"model.22.cv2.{lvl}.0", f)
            b = self.conv(f"model.22.cv2.{lvl}.1", b)
            b = self.conv(f"model.22.cv2.{lvl}.2", b, keep_fp32=True)
            c = self.conv(f"model.22.cv3.{lvl}.0", f)
            c = self.conv(f"model.22.cv3.{lvl}.1", c)
            c = self.conv(f"model.22.cv3.{lvl}.2", c, keep_fp32=True)
            h, w = f.shape[2], f.shape[3]
            boxes.append(b.reshape(B, 4 * self.reg_max, h * w))
            clss.append(c.reshape(B, self.nc, h * w))
            stride = (8, 16, 32)[lvl]
            sy, sx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5,
                                    torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
            anchors.append(torch.stack([sx.reshape(-1), sy.reshape(-1)], 0))   # [2, hw] (x, y)
            strides.append(torch.full((h * w,), float(stride)))
        box = torch.cat(boxes, 2)                                  # [B, 64, N]
        cls = torch.cat(clss, 2)                                   # [B, nc, N]
        anc = torch.cat(anchors, 1)[None]                          # [1, 2, N]
        st = torch.cat(strides)[None, None]                        # [1, 1, N]
        # DFL: softmax over reg_max bins, expectation
        N = box.shape[2]
        p = box.reshape(B, 4, self.reg_max, N).softmax(2)
        proj = torch.arange(self.reg_max, dtype=torch.float32).reshape(1, 1, -1, 1)
        dist = (p * proj).sum(2)                                   # [B, 4, N] = l, t, r, b
        lt, rb = dist[:, :2], dist[:, 2:]
        x1y1 = anc - lt
        x2y2 = anc + rb
        cxy = (x1y1 + x2y2) / 2
        wh = x2y2 - x1y1
        out_box = torch.cat([cxy, wh], 1) * st
        return torch.cat([out_box, cls.sigmoid()], 1)              # [B, 4+nc, N]


def load(path: str, mode: str = "fp32") -> YoloV8Ref:
    import importlib.util, os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location(
        "zly_model", os.path.join(here, "..", "zero-latency-yolo_amd", "tools", "zly_model.py"))
    zm = importlib.util.module_from_spec(spec)
    import sys
    sys.modules.setdefault("zly_model", zm)
    spec.loader.exec_module(zm)
    meta, weights = zm.read_zlyw(path)
    return YoloV8Ref(meta, weights, mode)
