"""YOLOv8 detect-model description, ZLYW weight-file I/O and the seeded synthetic-weight generator.

Host-side tooling for the HIP engine (not the oracle, and it imports nothing from oracle/).

The reference never describes the network: it hands an ONNX file produced by `ultralytics`
at install time to ONNX Runtime (reference start.sh:122-125, onnx_engine.cpp:578-585).  The
graph below is the public YOLOv8 `yolov8.yaml` detect architecture (SURVEY.md section 8c and
Appendix A), with every Conv = Conv2d(bias=False)+BatchNorm+SiLU already folded to
conv+bias+SiLU.  Conv names follow the ultralytics state-dict module paths so that a later
real-weights loader can fill the same file format.

ZLYW file (little endian), consumed by csrc/weights.cpp:
    magic "ZLYW", u32 version=1, u32 nc, u32 reg_max, u32 ch[5], u32 n_c2f[8], u32 num_convs,
    then num_convs records {char name[48]; u32 cin, cout, k, stride, act; u32 pad; u64 w_off, b_off},
    then fp32 payloads: weight [cout][cin][k][k] (PyTorch order) and bias [cout] per conv.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

MAGIC = b"ZLYW"
VERSION = 1
REC_FMT = "<48s5II2Q"  # name, cin,cout,k,stride,act, pad, w_off, b_off
REC_SIZE = struct.calcsize(REC_FMT)
HDR_FMT = "<4sIII5I8II"
HDR_SIZE = struct.calcsize(HDR_FMT)

# (width multiple, max channels, depth multiple) from the public yolov8.yaml scales
SCALES = {
    "n": (0.25, 1024, 1.0 / 3.0),
    "s": (0.50, 1024, 1.0 / 3.0),
    "m": (0.75, 768, 2.0 / 3.0),
}


@dataclass(frozen=True)
class ConvSpec:
    name: str
    cin: int
    cout: int
    k: int
    stride: int
    act: int  # 1 = SiLU, 0 = linear (the six final Detect convs)


@dataclass(frozen=True)
class ModelSpec:
    scale: str
    nc: int
    reg_max: int
    ch: Tuple[int, int, int, int, int]        # outputs of layers 0,1,3,5,7
    n_c2f: Tuple[int, ...]                    # bottleneck counts of C2f layers 2,4,6,8,12,15,18,21
    convs: Tuple[ConvSpec, ...]

    @property
    def head_c2(self) -> int:                 # Detect box-branch width
        return max(16, self.ch[2] // 4, self.reg_max * 4)

    @property
    def head_c3(self) -> int:                 # Detect class-branch width
        return max(self.ch[2], min(self.nc, 100))

    def num_anchors(self, w: int, h: int) -> int:
        return sum((h // s) * (w // s) for s in (8, 16, 32))

    def params(self) -> int:
        return sum(c.cout * c.cin * c.k * c.k + c.cout for c in self.convs)

    def macs(self, w: int, h: int) -> int:
        """multiply-accumulates per frame over all learnable convs."""
        total = 0
        for c, (ho, wo) in zip(self.convs, conv_output_sizes(self, w, h)):
            total += ho * wo * c.cout * c.cin * c.k * c.k
        return total


def _c2f(prefix: str, c1: int, c2: int, n: int) -> List[ConvSpec]:
    c = c2 // 2
    out = [ConvSpec(f"{prefix}.cv1", c1, 2 * c, 1, 1, 1)]
    for i in range(n):
        out.append(ConvSpec(f"{prefix}.m.{i}.cv1", c, c, 3, 1, 1))
        out.append(ConvSpec(f"{prefix}.m.{i}.cv2", c, c, 3, 1, 1))
    out.append(ConvSpec(f"{prefix}.cv2", (2 + n) * c, c2, 1, 1, 1))
    return out


def build_spec(scale: str = "n", nc: int = 80, reg_max: int = 16) -> ModelSpec:
    wm, cmax, dm = SCALES[scale]

    def width(c: int) -> int:
        v = min(c, cmax) * wm
        return int(-(-v // 8) * 8)            # make_divisible(x, 8) as ceil

    def depth(n: int) -> int:
        return max(round(n * dm), 1)

    ch = (width(64), width(128), width(256), width(512), width(1024))
    nb = (depth(3), depth(6), depth(6), depth(3))
    nn_ = depth(3)
    n_c2f = nb + (nn_, nn_, nn_, nn_)

    convs: List[ConvSpec] = []
    convs.append(ConvSpec("model.0", 3, ch[0], 3, 2, 1))
    convs.append(ConvSpec("model.1", ch[0], ch[1], 3, 2, 1))
    convs += _c2f("model.2", ch[1], ch[1], nb[0])
    convs.append(ConvSpec("model.3", ch[1], ch[2], 3, 2, 1))
    convs += _c2f("model.4", ch[2], ch[2], nb[1])
    convs.append(ConvSpec("model.5", ch[2], ch[3], 3, 2, 1))
    convs += _c2f("model.6", ch[3], ch[3], nb[2])
    convs.append(ConvSpec("model.7", ch[3], ch[4], 3, 2, 1))
    convs += _c2f("model.8", ch[4], ch[4], nb[3])
    convs.append(ConvSpec("model.9.cv1", ch[4], ch[4] // 2, 1, 1, 1))
    convs.append(ConvSpec("model.9.cv2", ch[4] * 2, ch[4], 1, 1, 1))
    convs += _c2f("model.12", ch[4] + ch[3], ch[3], nn_)
    convs += _c2f("model.15", ch[3] + ch[2], ch[2], nn_)
    convs.append(ConvSpec("model.16", ch[2], ch[2], 3, 2, 1))
    convs += _c2f("model.18", ch[2] + ch[3], ch[3], nn_)
    convs.append(ConvSpec("model.19", ch[3], ch[3], 3, 2, 1))
    convs += _c2f("model.21", ch[3] + ch[4], ch[4], nn_)
    c2 = max(16, ch[2] // 4, reg_max * 4)
    c3 = max(ch[2], min(nc, 100))
    for lvl, cin in enumerate((ch[2], ch[3], ch[4])):
        convs.append(ConvSpec(f"model.22.cv2.{lvl}.0", cin, c2, 3, 1, 1))
        convs.append(ConvSpec(f"model.22.cv2.{lvl}.1", c2, c2, 3, 1, 1))
        convs.append(ConvSpec(f"model.22.cv2.{lvl}.2", c2, 4 * reg_max, 1, 1, 0))
    for lvl, cin in enumerate((ch[2], ch[3], ch[4])):
        convs.append(ConvSpec(f"model.22.cv3.{lvl}.0", cin, c3, 3, 1, 1))
        convs.append(ConvSpec(f"model.22.cv3.{lvl}.1", c3, c3, 3, 1, 1))
        convs.append(ConvSpec(f"model.22.cv3.{lvl}.2", c3, nc, 1, 1, 0))
    return ModelSpec(scale, nc, reg_max, ch, n_c2f, tuple(convs))


def conv_output_sizes(spec: ModelSpec, w: int, h: int) -> List[Tuple[int, int]]:
    """(Ho, Wo) of every conv in spec order, for an input of w x h (must be a multiple of 32)."""
    lvl_of_layer = {0: 2, 1: 4, 2: 4, 3: 8, 4: 8, 5: 16, 6: 16, 7: 32, 8: 32, 9: 32,
                    12: 16, 15: 8, 16: 16, 18: 16, 19: 32, 21: 32}
    out = []
    for c in spec.convs:
        parts = c.name.split(".")
        layer = int(parts[1])
        if layer == 22:
            s = (8, 16, 32)[int(parts[3])]
        else:
            s = lvl_of_layer[layer]
        out.append((h // s, w // s))
    return out


# ----------------------------------------------------------------------------------------------
# file I/O
# ----------------------------------------------------------------------------------------------

def write_zlyw(path: str, spec: ModelSpec, weights: Dict[str, Tuple[np.ndarray, np.ndarray]]) -> None:
    recs = []
    off = HDR_SIZE + REC_SIZE * len(spec.convs)
    blobs = []
    for c in spec.convs:
        w, b = weights[c.name]
        assert w.shape == (c.cout, c.cin, c.k, c.k) and w.dtype == np.float32, (c.name, w.shape)
        assert b.shape == (c.cout,) and b.dtype == np.float32
        w_off = off
        off += w.nbytes
        b_off = off
        off += b.nbytes
        recs.append(struct.pack(REC_FMT, c.name.encode(), c.cin, c.cout, c.k, c.stride, c.act, 0, w_off, b_off))
        blobs.append(np.ascontiguousarray(w).tobytes())
        blobs.append(np.ascontiguousarray(b).tobytes())
    hdr = struct.pack(HDR_FMT, MAGIC, VERSION, spec.nc, spec.reg_max, *spec.ch, *spec.n_c2f, len(spec.convs))
    with open(path, "wb") as f:
        f.write(hdr)
        for r in recs:
            f.write(r)
        for b in blobs:
            f.write(b)


def read_zlyw(path: str):
    """-> (ModelSpec-like dict, {name: (w, b)}).  Used by tests and the torch oracle."""
    with open(path, "rb") as f:
        data = f.read()
    vals = struct.unpack_from(HDR_FMT, data, 0)
    if vals[0] != MAGIC or vals[1] != VERSION:
        raise ValueError("not a ZLYW v1 file: %s" % path)
    nc, reg_max = vals[2], vals[3]
    ch = tuple(vals[4:9])
    n_c2f = tuple(vals[9:17])
    num = vals[17]
    convs, weights = [], {}
    for i in range(num):
        name, cin, cout, k, stride, act, _pad, w_off, b_off = struct.unpack_from(REC_FMT, data, HDR_SIZE + i * REC_SIZE)
        name = name.rstrip(b"\0").decode()
        convs.append(ConvSpec(name, cin, cout, k, stride, act))
        w = np.frombuffer(data, dtype="<f4", count=cout * cin * k * k, offset=w_off).reshape(cout, cin, k, k)
        b = np.frombuffer(data, dtype="<f4", count=cout, offset=b_off)
        weights[name] = (w, b)
    meta = dict(nc=nc, reg_max=reg_max, ch=ch, n_c2f=n_c2f, convs=tuple(convs))
    return meta, weights


# ----------------------------------------------------------------------------------------------
# seeded synthetic weights (there are no real weights offline)
# ----------------------------------------------------------------------------------------------

# Real YOLOv8 weights carry folded BatchNorm statistics that keep every layer's activations O(1);
# plain He-style random weights do not (SiLU has no stable variance fixed point: activations grew
# to std ~140 by the head).  SYNTH_GAIN is the per-conv factor g in std = g / sqrt(fan_in) that
# makes the pre-activation std 1.0 (box logits 1.5) on the seeded 416x416 calibration frames.  The
# table was produced ONCE, offline, by oracle/calibrate_synth.py (LSUV-style, one forward pass) and
# is data here, so that the generator needs no forward pass and imports nothing from oracle/.
SYNTH_SEED = 8
BIAS_STD = 0.05
DFL_BIAS = 1.0            # box-branch final bias (ultralytics Detect.bias_init analogue)
CLS_LOGIT_SHIFT = -4.8    # class-branch final bias; puts ~1 % of anchors above conf 0.5 on noise frames
SYNTH_GAIN: Dict[str, float] = {
    "model.0": 3.6643,
    "model.1": 1.8027,
    "model.2.cv1": 1.8269,
    "model.2.m.0.cv1": 1.8764,
    "model.2.m.0.cv2": 1.7797,
    "model.2.cv2": 1.5978,
    "model.3": 1.7415,
    "model.4.cv1": 1.8730,
    "model.4.m.0.cv1": 1.8319,
    "model.4.m.0.cv2": 1.7611,
    "model.4.m.1.cv1": 1.2753,
    "model.4.m.1.cv2": 1.7753,
    "model.4.cv2": 1.3218,
    "model.5": 1.7712,
    "model.6.cv1": 1.7718,
    "model.6.m.0.cv1": 1.7174,
    "model.6.m.0.cv2": 1.8384,
    "model.6.m.1.cv1": 1.2422,
    "model.6.m.1.cv2": 1.8498,
    "model.6.cv2": 1.2983,
    "model.7": 1.8381,
    "model.8.cv1": 1.7790,
    "model.8.m.0.cv1": 1.7903,
    "model.8.m.0.cv2": 1.8016,
    "model.8.cv2": 1.5082,
    "model.9.cv1": 1.7939,
    "model.9.cv2": 0.8837,
    "model.12.cv1": 1.8244,
    "model.12.m.0.cv1": 1.6148,
    "model.12.m.0.cv2": 1.9027,
    "model.12.cv2": 1.7182,
    "model.15.cv1": 1.8125,
    "model.15.m.0.cv1": 1.9693,
    "model.15.m.0.cv2": 1.8546,
    "model.15.cv2": 1.7808,
    "model.16": 1.8890,
    "model.18.cv1": 1.8442,
    "model.18.m.0.cv1": 1.8129,
    "model.18.m.0.cv2": 1.8503,
    "model.18.cv2": 1.8183,
    "model.19": 1.7555,
    "model.21.cv1": 1.8484,
    "model.21.m.0.cv1": 1.7577,
    "model.21.m.0.cv2": 1.7880,
    "model.21.cv2": 1.8006,
    "model.22.cv2.0.0": 1.8828,
    "model.22.cv2.0.1": 1.7913,
    "model.22.cv2.0.2": 2.6417,
    "model.22.cv2.1.0": 1.7405,
    "model.22.cv2.1.1": 1.8234,
    "model.22.cv2.1.2": 2.6174,
    "model.22.cv2.2.0": 1.8975,
    "model.22.cv2.2.1": 1.8689,
    "model.22.cv2.2.2": 2.7418,
    "model.22.cv3.0.0": 1.8901,
    "model.22.cv3.0.1": 1.8143,
    "model.22.cv3.0.2": 1.8162,
    "model.22.cv3.1.0": 1.7667,
    "model.22.cv3.1.1": 1.8474,
    "model.22.cv3.1.2": 1.7116,
    "model.22.cv3.2.0": 1.7800,
    "model.22.cv3.2.1": 1.8568,
    "model.22.cv3.2.2": 1.9786,
}


def synth_weights(spec: ModelSpec, seed: int = SYNTH_SEED, gains: Dict[str, float] = None) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    gains = SYNTH_GAIN if gains is None else gains
    rng = np.random.default_rng(seed)
    out = {}
    for c in spec.convs:
        fan_in = c.cin * c.k * c.k
        gain = gains.get(c.name, 1.0)
        w = rng.standard_normal((c.cout, c.cin, c.k, c.k), dtype=np.float32)
        # zero-mean filters: SiLU outputs are mostly positive, and a filter with a non-zero mean turns
        # that common mode into a frame-wide logit offset (whole frames above or below threshold)
        w -= w.mean(axis=(1, 2, 3), keepdims=True)
        w *= np.float32(gain / np.sqrt(fan_in))
        b = rng.standard_normal((c.cout,), dtype=np.float32) * np.float32(BIAS_STD)
        if c.name.startswith("model.22.cv2.") and c.name.endswith(".2"):
            b = b + np.float32(DFL_BIAS)
        if c.name.startswith("model.22.cv3.") and c.name.endswith(".2"):
            b = b + np.float32(CLS_LOGIT_SHIFT)
        out[c.name] = (w.astype(np.float32), b.astype(np.float32))
    return out


def synth_frames(n: int, w: int, h: int, seed: int = 20250328, rects: bool = True) -> np.ndarray:
    """Seeded synthetic u8 BGR frames [n][h][w][3] (SURVEY.md section 8d 'Synthetic inputs').
    rects=False: uniform noise, statistically identical frame to frame (bench / calibration set).
    rects=True : smooth gradient background + 4-12 pasted constant-colour rectangles + mild noise."""
    rng = np.random.default_rng(seed)
    if not rects:
        return rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    frames = np.empty((n, h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for i in range(n):
        base = rng.integers(0, 256, size=3).astype(np.float32)
        gx, gy = rng.uniform(-0.4, 0.4, size=2).astype(np.float32)
        img = base[None, None, :] + gx * xx[..., None] + gy * yy[..., None]
        for _ in range(int(rng.integers(4, 13))):
            x0, y0 = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
            rw, rh = int(rng.integers(8, max(9, w // 2))), int(rng.integers(8, max(9, h // 2)))
            img[y0:y0 + rh, x0:x0 + rw, :] = rng.integers(0, 256, size=3).astype(np.float32)
        img = img + rng.normal(0.0, 6.0, size=img.shape).astype(np.float32)
        frames[i] = np.clip(img, 0, 255).astype(np.uint8)
    return frames


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(description="write a seeded synthetic ZLYW weight file")
    ap.add_argument("--scale", default="n", choices=sorted(SCALES))
    ap.add_argument("--nc", type=int, default=80)
    ap.add_argument("--seed", type=int, default=SYNTH_SEED)
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args(argv)
    spec = build_spec(a.scale, a.nc)
    write_zlyw(a.out, spec, synth_weights(spec, a.seed))
    print(f"{a.out}: yolov8{a.scale} nc={a.nc} convs={len(spec.convs)} params={spec.params()}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
