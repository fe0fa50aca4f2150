"""Test helper: writes a small but well-formed ONNX ModelProto (protobuf wire format by hand) holding Conv nodes and
their initializers, the way an exporter lays out a fused YOLOv8: used to exercise tools/onnx_min.py and
tools/convert_weights.py.  Field numbers: published onnx.proto3 (see tools/onnx_min.py)."""
import struct

import numpy as np


def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fno: int, payload: bytes) -> bytes:
    return _varint(fno << 3 | 2) + _varint(len(payload)) + payload


def _vi(fno: int, v: int) -> bytes:
    return _varint(fno << 3 | 0) + _varint(v)


def tensor(name: str, arr: np.ndarray, style: str = "raw") -> bytes:
    arr = np.asarray(arr)
    dt = {np.dtype("float32"): 1, np.dtype("int64"): 7, np.dtype("float16"): 10}[arr.dtype]
    if style == "packed_dims":
        t = _ld(1, b"".join(_varint(d) for d in arr.shape))
    else:
        t = b"".join(_vi(1, d) for d in arr.shape)
    t += _vi(2, dt)
    if style == "float_data" and dt == 1:
        t += _ld(4, arr.astype("<f4").tobytes())
    elif style == "int64_data" and dt == 7:
        t += _ld(7, b"".join(_varint(int(x)) for x in arr.reshape(-1)))
    else:
        t += _ld(9, arr.astype(arr.dtype.newbyteorder("<")).tobytes())
    return t + _ld(8, name.encode())


def node(op: str, inputs, outputs, name: str = "") -> bytes:
    n = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    if name:
        n += _ld(3, name.encode())
    return n + _ld(4, op.encode())


def model(nodes, initializers) -> bytes:
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"main_graph") + b"".join(_ld(5, t) for t in initializers)
    opset = _ld(1, b"") + _vi(2, 17)
    return _vi(1, 8) + _ld(2, b"pytorch") + _ld(3, b"2.x") + _ld(7, graph) + _ld(8, opset)
