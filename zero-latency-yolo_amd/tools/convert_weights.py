"""Real-weights loader: YOLOv8 detection weights -> the engine's flat ZLYW file.

Inputs (no ultralytics / onnx package needed):
  --state-dict FILE   a torch.save()d flat state dict {module path + '.conv.weight' / '.bn.*' / '.weight' / '.bias': tensor}
                      as `DetectionModel.state_dict()` gives it (dump it once where ultralytics is installed:
                      `torch.save(YOLO('yolov8n.pt').model.state_dict(), 'yolov8n_sd.pt')`); loaded with
                      weights_only=True.  Conv+BatchNorm pairs are folded here (eps 1e-3, ultralytics' BatchNorm2d setting).
  --onnx FILE         the file `yolo export model=yolov8n.pt format=onnx` writes (reference start.sh:122-125), the model
                      the reference's OnnxInferenceEngine loads (onnx_engine.cpp:984-1040): BN is already fused by the
                      exporter; convs are matched by initializer name ('model.N...conv.weight'), or, if the names were
                      rewritten, by Conv-node order against the module order with every shape checked.
The variant (n/s/m/l/x), class count (80 for COCO, 4 for the reference's CS 1.6 head, constants.h:35-40) and reg_max are
read off the tensor shapes.  The DFL projection must be arange(reg_max): the engine computes the expectation in-kernel.

The reference ships neither weights nor an .onnx, so no real file has been through this converter here: PARITY UNPINNED
against a real export; tests cover the BN folding against torch, the shape/variant inference and both container formats."""
import argparse
import re
import sys
from typing import Dict, Tuple

import numpy as np

import zly_model as zm

BN_EPS = 1e-3
_WIDTH = {16: "n", 32: "s", 48: "m", 64: "l", 80: "x"}


def fold_bn(w: np.ndarray, gamma, beta, mean, var, eps: float = BN_EPS) -> Tuple[np.ndarray, np.ndarray]:
    """conv (no bias) followed by eval-mode BatchNorm == conv with w' = w * g/sqrt(var+eps), b' = beta - mean * g/sqrt(var+eps)"""
    scale = (gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + eps))
    return (w.astype(np.float64) * scale[:, None, None, None]).astype(np.float32), (beta.astype(np.float64) - mean.astype(np.float64) * scale).astype(np.float32)


def infer_spec(shape_of) -> zm.ModelSpec:
    """shape_of(conv name) -> weight shape; picks the variant whose every conv shape matches"""
    c0 = shape_of("model.0")[0]
    if c0 not in _WIDTH:
        raise ValueError(f"model.0 has {c0} output channels: not a YOLOv8 n/s/m/l/x detection model")
    nc = shape_of("model.22.cv3.0.2")[0]
    reg4 = shape_of("model.22.cv2.0.2")[0]
    if reg4 % 4:
        raise ValueError(f"box branch has {reg4} outputs, not 4*reg_max")
    spec = zm.build_spec(_WIDTH[c0], nc, reg4 // 4)
    for c in spec.convs:
        got = tuple(shape_of(c.name))
        if got != (c.cout, c.cin, c.k, c.k):
            raise ValueError(f"{c.name}: weight shape {got}, expected {(c.cout, c.cin, c.k, c.k)} for yolov8{spec.scale}")
    return spec


def _check_dfl(w, reg_max):
    if w is not None and not np.array_equal(np.asarray(w, dtype=np.float32).reshape(-1), np.arange(reg_max, dtype=np.float32)):
        raise ValueError("model.22.dfl.conv.weight is not arange(reg_max): unsupported DFL projection")


def from_state_dict(sd: Dict[str, np.ndarray]):
    sd = {k: np.asarray(v, dtype=np.float32) if np.asarray(v).dtype.kind == "f" else np.asarray(v) for k, v in sd.items()}

    def wkey(name):
        for k in (name + ".conv.weight", name + ".weight"):
            if k in sd:
                return k
        raise KeyError(f"no weight for {name} ({name}.conv.weight / {name}.weight)")

    spec = infer_spec(lambda n: sd[wkey(n)].shape)
    out = {}
    for c in spec.convs:
        k = wkey(c.name)
        w = sd[k]
        if k.endswith(".conv.weight") and c.name + ".bn.weight" in sd:
            p = c.name + ".bn."
            w, b = fold_bn(w, sd[p + "weight"], sd[p + "bias"], sd[p + "running_mean"], sd[p + "running_var"])
        else:
            bk = k[:-len("weight")] + "bias"
            b = sd[bk] if bk in sd else np.zeros(c.cout, np.float32)     # a fused checkpoint keeps the bias on .conv
        out[c.name] = (np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32))
    _check_dfl(sd.get("model.22.dfl.conv.weight"), spec.reg_max)
    return spec, out


def from_onnx(path: str):
    import onnx_min
    inits, nodes = onnx_min.read_onnx(path)
    named = {}
    for k, v in inits.items():
        m = re.match(r"^(model\.\d+(?:\.[\w]+)*?)(?:\.conv)?\.weight$", k)
        if m and v.ndim == 4 and not m.group(1).endswith(".dfl"):
            named[m.group(1)] = (k, k[:-len("weight")] + "bias")
    if "model.0" in named and "model.22.cv3.0.2" in named:
        spec = infer_spec(lambda n: inits[named[n][0]].shape)
        pairs = {c.name: named[c.name] for c in spec.convs}
    else:                                           # names rewritten by a graph optimiser: match Conv nodes in module order
        convs = [n for n in nodes if n["op_type"] == "Conv" and len(n["inputs"]) >= 2 and n["inputs"][1] in inits and inits[n["inputs"][1]].ndim == 4]
        convs = [n for n in convs if not (inits[n["inputs"][1]].shape[0] == 1 and inits[n["inputs"][1]].shape[1] <= 32)]   # drop the DFL conv [1, reg_max, 1, 1]
        order = lambda spec: [c.name for c in spec.convs if not c.name.startswith("model.22.")] + \
            [f"model.22.cv{b}.{l}.{i}" for l in range(3) for b in (2, 3) for i in range(3)]      # Detect.forward: cv2[l] then cv3[l], per level
        spec0 = zm.build_spec(_WIDTH.get(inits[convs[0]["inputs"][1]].shape[0], "n"))
        if len(convs) != len(spec0.convs):
            raise ValueError(f"{path}: {len(convs)} Conv nodes with weights, expected {len(spec0.convs)}")
        by_order = dict(zip(order(spec0), convs))
        spec = infer_spec(lambda n: inits[by_order[n]["inputs"][1]].shape)
        pairs = {n: (by_order[n]["inputs"][1], by_order[n]["inputs"][2] if len(by_order[n]["inputs"]) > 2 else "") for n in by_order}
    out = {}
    for c in spec.convs:
        wk, bk = pairs[c.name]
        b = inits[bk] if bk in inits else np.zeros(c.cout, np.float32)
        out[c.name] = (np.ascontiguousarray(inits[wk], np.float32), np.ascontiguousarray(b, np.float32).reshape(-1))
    _check_dfl(inits.get("model.22.dfl.conv.weight"), spec.reg_max)
    return spec, out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--state-dict")
    g.add_argument("--onnx")
    ap.add_argument("--out", required=True)
    a = ap.parse_args(argv)
    if a.state_dict:
        import torch
        sd = torch.load(a.state_dict, map_location="cpu", weights_only=True)
        if not isinstance(sd, dict) or not all(hasattr(v, "numpy") for v in sd.values()):
            print("expected a flat state dict of tensors (see --help)", file=sys.stderr)
            return 2
        spec, w = from_state_dict({k: v.float().numpy() if v.is_floating_point() else v.numpy() for k, v in sd.items()})
    else:
        spec, w = from_onnx(a.onnx)
    zm.write_zlyw(a.out, spec, w)
    print(f"{a.out}: yolov8{spec.scale} nc={spec.nc} reg_max={spec.reg_max} convs={len(spec.convs)} params={spec.params()}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
