"""Minimal ONNX initializer reader: protobuf wire format only, no `onnx` / `protobuf` package.

Reads what a weight converter needs from an ONNX ModelProto: the graph's initializers (name, dims, data) and its
nodes (op_type, inputs, outputs) in file order.  Field numbers follow the published onnx.proto3:
  ModelProto.graph = 7;  GraphProto.node = 1, .initializer = 5;
  NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4;
  TensorProto.dims = 1, .data_type = 2, .float_data = 4, .int64_data = 7, .name = 8, .raw_data = 9, .data_location = 14
  TensorProto.DataType: FLOAT = 1, INT64 = 7, FLOAT16 = 10.
The reference exports its model with `yolo export format=onnx` at install time (reference start.sh:122-125) and ships no
.onnx file, so this reader is checked against files written by tests/onnx_writer.py only: PARITY UNPINNED against a
real export."""
import struct
from typing import Dict, Iterator, List, Tuple

import numpy as np

_DTYPES = {1: np.float32, 7: np.int64, 10: np.float16, 11: np.float64, 6: np.int32}


def _varint(buf: memoryview, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("varint too long")


def fields(buf: memoryview) -> Iterator[Tuple[int, int, object]]:
    """yields (field number, wire type, value): int for varint/fixed, memoryview for length-delimited"""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            if pos + n > end:
                raise ValueError("truncated length-delimited field")
            val = buf[pos:pos + n]; pos += n
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, val


def _tensor(buf: memoryview) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype, name, raw, floats, int64s, location = 0, "", None, [], [], 0
    for fno, wt, val in fields(buf):
        if fno == 1:
            if wt == 2:                                    # packed
                p = 0
                while p < len(val):
                    d, p = _varint(val, p); dims.append(d)
            else:
                dims.append(val)
        elif fno == 2: dtype = val
        elif fno == 4:
            floats.append(np.frombuffer(val, dtype="<f4") if wt == 2 else np.array([struct.unpack("<f", struct.pack("<I", val))[0]], dtype=np.float32))
        elif fno == 7:
            if wt == 2:
                p = 0
                while p < len(val):
                    d, p = _varint(val, p); int64s.append(d - (1 << 64) if d >> 63 else d)
            else:
                int64s.append(val - (1 << 64) if val >> 63 else val)
        elif fno == 8: name = bytes(val).decode()
        elif fno == 9: raw = val
        elif fno == 14: location = val
    if location == 1:
        raise ValueError(f"initializer {name}: external data is not supported")
    if dtype not in _DTYPES:
        raise ValueError(f"initializer {name}: unsupported data_type {dtype}")
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np.dtype(_DTYPES[dtype]).newbyteorder("<"))
    elif floats:
        arr = np.concatenate(floats)
    elif int64s:
        arr = np.array(int64s, dtype=np.int64)
    else:
        arr = np.zeros(0, dtype=_DTYPES[dtype])
    n = int(np.prod(dims)) if dims else arr.size
    if arr.size != n:
        raise ValueError(f"initializer {name}: {arr.size} elements for dims {dims}")
    return name, arr.reshape(dims).astype(_DTYPES[dtype], copy=False)


def _node(buf: memoryview) -> dict:
    n = dict(op_type="", name="", inputs=[], outputs=[])
    for fno, _wt, val in fields(buf):
        if fno == 1: n["inputs"].append(bytes(val).decode())
        elif fno == 2: n["outputs"].append(bytes(val).decode())
        elif fno == 3: n["name"] = bytes(val).decode()
        elif fno == 4: n["op_type"] = bytes(val).decode()
    return n


def read_onnx(path: str) -> Tuple[Dict[str, np.ndarray], List[dict]]:
    """-> ({initializer name: array}, [node dicts in graph order])"""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    graph = None
    for fno, wt, val in fields(buf):
        if fno == 7 and wt == 2:
            graph = val
    if graph is None:
        raise ValueError(f"{path}: no GraphProto (not an ONNX model?)")
    inits: Dict[str, np.ndarray] = {}
    nodes: List[dict] = []
    for fno, wt, val in fields(graph):
        if fno == 5 and wt == 2:
            name, arr = _tensor(val)
            inits[name] = arr
        elif fno == 1 and wt == 2:
            nodes.append(_node(val))
    return inits, nodes
