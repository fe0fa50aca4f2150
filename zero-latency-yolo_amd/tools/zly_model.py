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
    then num_convs records {char name[48]; u32 cin, cout, k, stride, act; u32 wfmt; u64 w_off, b_off},
    then payloads per conv: weight [cout][cin][k][k] (PyTorch order) as fp32 (wfmt 0) or as OCP fp8 e4m3 bytes preceded by one int8
    power-of-two exponent per output channel, padded to 4 bytes (wfmt 1: w = e4m3 * 2^exp[cout]); bias [cout] fp32.
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

def _e4m3_table() -> np.ndarray:
    """the 256 OCP fp8 e4m3 (e4m3fn) values, NaN at 0x7f / 0xff"""
    v = np.arange(256, dtype=np.uint32)
    ex, man = (v >> 3) & 15, v & 7
    mag = np.where(ex == 0, man / 8.0 * 2.0 ** -6, (1.0 + man / 8.0) * 2.0 ** (ex.astype(np.float64) - 7))
    mag = np.where((ex == 15) & (man == 7), np.nan, mag)
    return np.where(v >> 7 == 1, -mag, mag).astype(np.float32)


E4M3 = _e4m3_table()
E4M3_MAX = 448.0


def quantize_fp8(w: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """fp32 [cout][...] -> (int8 exponent per output channel, uint8 e4m3 codes): w ~ e4m3 * 2^exp, the exponent chosen so that the
    channel's largest weight lands in [224, 448]; round to nearest, ties to even code (fp8 weights, BASELINE configs[4])."""
    flat = w.reshape(w.shape[0], -1).astype(np.float64)
    amax = np.abs(flat).max(1)
    ex = np.where(amax > 0, np.ceil(np.log2(np.maximum(amax, 1e-30) / E4M3_MAX)), 0).astype(np.int64)
    ex = np.clip(ex, -126, 126)
    x = flat / (2.0 ** ex)[:, None]
    pos = E4M3[:127].astype(np.float64)                        # codes 0x00..0x7e ascending
    a = np.minimum(np.abs(x), E4M3_MAX)
    hi = np.clip(np.searchsorted(pos, a, side="left"), 1, 126)
    lo = hi - 1
    dlo, dhi = a - pos[lo], pos[hi] - a
    code = np.where((dhi < dlo) | ((dhi == dlo) & (hi % 2 == 0)), hi, lo).astype(np.uint8)
    code = np.where(x < 0, code | 0x80, code).astype(np.uint8)
    return ex.astype(np.int8), code.reshape(w.shape)


def dequantize_fp8(ex: np.ndarray, code: np.ndarray) -> np.ndarray:
    return (E4M3[code].reshape(code.shape[0], -1) * (2.0 ** ex.astype(np.float64))[:, None].astype(np.float32)).reshape(code.shape).astype(np.float32)


def write_zlyw(path: str, spec: ModelSpec, weights: Dict[str, Tuple[np.ndarray, np.ndarray]], fp8: bool = False) -> None:
    """fp8 = True: every conv's weights are stored as e4m3 codes + per-output-channel power-of-two exponents (1 byte per weight)."""
    recs = []
    off = HDR_SIZE + REC_SIZE * len(spec.convs)
    blobs = []
    for c in spec.convs:
        w, b = weights[c.name]
        assert w.shape == (c.cout, c.cin, c.k, c.k) and w.dtype == np.float32, (c.name, w.shape)
        assert b.shape == (c.cout,) and b.dtype == np.float32
        if fp8:
            ex, code = quantize_fp8(w)
            wblob = ex.tobytes() + b"\0" * ((-len(ex)) % 4) + np.ascontiguousarray(code).tobytes()
        else:
            wblob = np.ascontiguousarray(w).tobytes()
        wblob += b"\0" * ((-len(wblob)) % 4)
        w_off = off
        off += len(wblob)
        b_off = off
        off += b.nbytes
        recs.append(struct.pack(REC_FMT, c.name.encode(), c.cin, c.cout, c.k, c.stride, c.act, 1 if fp8 else 0, w_off, b_off))
        blobs.append(wblob)
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
        name, cin, cout, k, stride, act, wfmt, w_off, b_off = struct.unpack_from(REC_FMT, data, HDR_SIZE + i * REC_SIZE)
        name = name.rstrip(b"\0").decode()
        convs.append(ConvSpec(name, cin, cout, k, stride, act))
        if wfmt == 1:
            ex = np.frombuffer(data, dtype=np.int8, count=cout, offset=w_off)
            code = np.frombuffer(data, dtype=np.uint8, count=cout * cin * k * k, offset=w_off + (cout + 3) // 4 * 4).reshape(cout, cin, k, k)
            w = dequantize_fp8(ex, code)
        else:
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
# makes the pre-activation std SYNTH_ACT_STD on the seeded 416x416 calibration frames.  The
# table was produced ONCE, offline, by oracle/calibrate_synth.py (LSUV-style, one forward pass) and
# is data here, so that the generator needs no forward pass and imports nothing from oracle/.
#
# Two properties of a TRAINED detector are built in, because the bf16 tolerance of SURVEY.md 8c
# (box <= 1.5 px, score <= 2e-2) is only meaningful on a network that has them:
#  * noise stability.  A random deep net calibrated to pre-activation std 1.0 is in the chaotic regime:
#    a perturbation grows ~1.2x per layer, so the 2^-9 rounding noise of bf16 activations reached 7 % of
#    the signal at the Detect head (measured on the CPU: bf16-rounding oracle vs fp32 oracle, round 1).
#    At std 0.25 SiLU works on its near-linear part, perturbations neither grow nor shrink, and the
#    head sees ~1 % = 2^-9 * sqrt(depth), which is what bf16 inference of a trained model shows.
#  * peaked DFL distributions.  The final box conv (cv2.L.2) emits, per box side, 16 logits
#    z_i = -a*i^2 + 2a*i*mu  (+ a term constant in i), i.e. softmax_i = a discretised Gaussian of
#    mean mu and sigma = 1/sqrt(2a) bins, where mu = w_mu . x + b_mu is LINEAR in the features: row i
#    of the conv is 2a*i*w_mu, bias -a*i^2 + 2a*i*b_mu.  With a = 0.5 (sigma = 1 bin) the DFL
#    expectation is mu itself, as in a trained head that puts its mass on the two bins around the
#    regressed distance -- not the near-uniform softmax of random logits whose expectation amplifies
#    logit noise by several bins.
SYNTH_SEED = 8
BIAS_STD = 0.05
SYNTH_ACT_STD = 0.25      # pre-activation std of every SiLU conv (calibration target)
DFL_ALPHA = 0.5           # Gaussian DFL: sigma = 1 bin
DFL_MU_BIAS = 4.0         # mean regressed distance in bins (boxes ~8 bins = 64 / 128 / 256 px wide)
DFL_MU_STD = 0.3          # std of the regressed distance over anchors (calibration target, bins)
CLS_LOGIT_STD = 0.7       # std of the class logits (calibration target)
CLS_LOGIT_SHIFT = -2.75   # class-branch final bias; puts ~1 % of anchors above conf 0.5 on noise frames
SYNTH_GAIN: Dict[str, float] = {
    "model.0": 0.9161,
    "model.1": 1.9635,
    "model.2.cv1": 2.0015,
    "model.2.m.0.cv1": 2.0066,
    "model.2.m.0.cv2": 1.8224,
    "model.2.cv2": 1.7142,
    "model.3": 1.8793,
    "model.4.cv1": 2.1230,
    "model.4.m.0.cv1": 1.9933,
    "model.4.m.0.cv2": 2.0012,
    "model.4.m.1.cv1": 1.3976,
    "model.4.m.1.cv2": 1.8836,
    "model.4.cv2": 1.4595,
    "model.5": 1.9976,
    "model.6.cv1": 1.8835,
    "model.6.m.0.cv1": 1.8026,
    "model.6.m.0.cv2": 2.2138,
    "model.6.m.1.cv1": 1.3479,
    "model.6.m.1.cv2": 2.0085,
    "model.6.cv2": 1.3720,
    "model.7": 1.9759,
    "model.8.cv1": 1.9129,
    "model.8.m.0.cv1": 1.8854,
    "model.8.m.0.cv2": 2.0056,
    "model.8.cv2": 1.6757,
    "model.9.cv1": 1.8604,
    "model.9.cv2": 1.3642,
    "model.12.cv1": 2.0366,
    "model.12.m.0.cv1": 1.8865,
    "model.12.m.0.cv2": 2.0101,
    "model.12.cv2": 1.9545,
    "model.15.cv1": 2.0030,
    "model.15.m.0.cv1": 2.2408,
    "model.15.m.0.cv2": 1.9484,
    "model.15.cv2": 1.8297,
    "model.16": 1.9649,
    "model.18.cv1": 1.9535,
    "model.18.m.0.cv1": 1.9982,
    "model.18.m.0.cv2": 2.1151,
    "model.18.cv2": 1.9064,
    "model.19": 1.8504,
    "model.21.cv1": 2.0813,
    "model.21.m.0.cv1": 2.1656,
    "model.21.m.0.cv2": 1.9412,
    "model.21.cv2": 1.9640,
    "model.22.cv2.0.0": 1.9791,
    "model.22.cv2.0.1": 1.9290,
    "model.22.cv2.0.2": 2.4489,
    "model.22.cv2.1.0": 1.8420,
    "model.22.cv2.1.1": 1.9390,
    "model.22.cv2.1.2": 2.9245,
    "model.22.cv2.2.0": 1.9841,
    "model.22.cv2.2.1": 1.9732,
    "model.22.cv2.2.2": 2.4386,
    "model.22.cv3.0.0": 1.9352,
    "model.22.cv3.0.1": 1.9841,
    "model.22.cv3.0.2": 6.0178,
    "model.22.cv3.1.0": 1.7576,
    "model.22.cv3.1.1": 2.1012,
    "model.22.cv3.1.2": 5.3266,
    "model.22.cv3.2.0": 2.0347,
    "model.22.cv3.2.1": 2.0770,
    "model.22.cv3.2.2": 5.9230,
}

SYNTH_GAIN_S: Dict[str, float] = {   # yolov8s (python oracle/calibrate_synth.py s)
    "model.0": 0.9219,
    "model.1": 1.9173,
    "model.2.cv1": 1.9658,
    "model.2.m.0.cv1": 1.8716,
    "model.2.m.0.cv2": 1.9020,
    "model.2.cv2": 1.6942,
    "model.3": 1.9436,
    "model.4.cv1": 1.8790,
    "model.4.m.0.cv1": 1.9985,
    "model.4.m.0.cv2": 1.8751,
    "model.4.m.1.cv1": 1.4003,
    "model.4.m.1.cv2": 2.0557,
    "model.4.cv2": 1.4371,
    "model.5": 2.0193,
    "model.6.cv1": 1.9267,
    "model.6.m.0.cv1": 1.9631,
    "model.6.m.0.cv2": 1.9377,
    "model.6.m.1.cv1": 1.3799,
    "model.6.m.1.cv2": 2.0010,
    "model.6.cv2": 1.4404,
    "model.7": 1.9913,
    "model.8.cv1": 1.9497,
    "model.8.m.0.cv1": 1.9548,
    "model.8.m.0.cv2": 1.9475,
    "model.8.cv2": 1.6925,
    "model.9.cv1": 1.9220,
    "model.9.cv2": 1.3687,
    "model.12.cv1": 1.9651,
    "model.12.m.0.cv1": 1.8716,
    "model.12.m.0.cv2": 1.7766,
    "model.12.cv2": 1.9189,
    "model.15.cv1": 1.9516,
    "model.15.m.0.cv1": 2.0208,
    "model.15.m.0.cv2": 2.1398,
    "model.15.cv2": 2.0182,
    "model.16": 1.8264,
    "model.18.cv1": 1.9092,
    "model.18.m.0.cv1": 1.8121,
    "model.18.m.0.cv2": 1.8984,
    "model.18.cv2": 1.9352,
    "model.19": 1.9111,
    "model.21.cv1": 1.9716,
    "model.21.m.0.cv1": 1.8321,
    "model.21.m.0.cv2": 2.0457,
    "model.21.cv2": 1.9123,
    "model.22.cv2.0.0": 1.9744,
    "model.22.cv2.0.1": 1.7448,
    "model.22.cv2.0.2": 2.7074,
    "model.22.cv2.1.0": 2.0395,
    "model.22.cv2.1.1": 1.9996,
    "model.22.cv2.1.2": 2.9802,
    "model.22.cv2.2.0": 2.2310,
    "model.22.cv2.2.1": 2.0006,
    "model.22.cv2.2.2": 4.7547,
    "model.22.cv3.0.0": 2.0549,
    "model.22.cv3.0.1": 1.9239,
    "model.22.cv3.0.2": 5.3305,
    "model.22.cv3.1.0": 1.9880,
    "model.22.cv3.1.1": 2.0467,
    "model.22.cv3.1.2": 5.9566,
    "model.22.cv3.2.0": 2.0336,
    "model.22.cv3.2.1": 1.9636,
    "model.22.cv3.2.2": 5.4035,
}


def synth_weights(spec: ModelSpec, seed: int = SYNTH_SEED, gains: Dict[str, float] = None) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    gains = (SYNTH_GAIN_S if spec.scale == "s" else SYNTH_GAIN) if gains is None else gains
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
            # Gaussian DFL head (see above): side s uses filter row 16*s as w_mu and its bias (+ DFL_MU_BIAS) as b_mu
            assert c.cout == 4 * spec.reg_max and c.k == 1
            bins = np.arange(spec.reg_max, dtype=np.float32)
            w_mu = w[::spec.reg_max].copy()                                  # [4][cin][1][1]
            b_mu = b[::spec.reg_max].copy() + np.float32(DFL_MU_BIAS)        # [4]
            two_a = np.float32(2.0 * DFL_ALPHA)
            w = (two_a * bins[None, :, None, None, None] * w_mu[:, None]).reshape(c.cout, c.cin, 1, 1)
            b = (-np.float32(DFL_ALPHA) * bins[None, :] ** 2 + two_a * bins[None, :] * b_mu[:, None]).reshape(c.cout)
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
    ap.add_argument("--fp8", action="store_true", help="store the weights as fp8 e4m3 + per-output-channel power-of-two exponents")
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args(argv)
    spec = build_spec(a.scale, a.nc)
    write_zlyw(a.out, spec, synth_weights(spec, a.seed), fp8=a.fp8)
    print(f"{a.out}: yolov8{a.scale} nc={a.nc} convs={len(spec.convs)} params={spec.params()}" + (" fp8-e4m3 weights" if a.fp8 else ""))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
