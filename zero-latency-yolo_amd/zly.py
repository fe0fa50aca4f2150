"""ctypes binding of include/zly.h (libzly.so) for tests, bench.py and smoke().

Plumbing only: every call goes straight through the C ABI.  There is no Python or CPU fallback:
if the HIP extension is missing, or no HIP device is visible, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libzly.so")
DEFAULT_WEIGHTS = os.path.join(_HERE, "_build", "yolov8n_synth.zlyw")

OK = 0
ERR_INVALID_ARGUMENT = 2
ERR_NOT_INITIALIZED = 3
ERR_INFERENCE = 200
ERR_MODEL_NOT_FOUND = 201
ERR_MODEL_LOAD = 202
ERR_INVALID_INPUT = 203
ERR_SYSTEM = 300
PENDING = 1
DTYPE_FP32, DTYPE_BF16 = 0, 1
SLAB_OVERFLOW = 1
FLAG_DUMP_LOGITS = 1
FLAG_NO_FUSION = 2
FLAG_NO_HEAD_TENSOR = 4
FLAG_ASYNC_NMS = 8
FLAG_SINGLE_CHAIN = 16

# every symbol include/zly.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = [
    "zly_default_config", "zly_create", "zly_destroy", "zly_last_error", "zly_version",
    "zly_detect", "zly_detect_batch", "zly_submit", "zly_submit_try", "zly_poll", "zly_wait", "zly_detect_device", "zly_slab_bytes", "zly_read_slabs", "zly_sync", "zly_join",
    "zly_preprocess", "zly_forward", "zly_head_tensor", "zly_postprocess", "zly_debug_tap",
    "zly_num_classes", "zly_weights_fp8", "zly_num_anchors", "zly_num_ops", "zly_op_info_at", "zly_launch_info_at", "zly_op_kernel_name", "zly_profile_ops", "zly_get_stats",
]

DET_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("w", "<f4"), ("h", "<f4"), ("confidence", "<f4"),
                      ("class_id", "<i4"), ("track_id", "<u4"), ("pad_", "<u4"), ("timestamp", "<u8")])
assert DET_DTYPE.itemsize == 40
SLAB_HDR_DTYPE = np.dtype([("n_kept", "<i4"), ("n_candidates", "<i4"), ("flags", "<u4"), ("frame_tag", "<u4")])


class Config(C.Structure):
    _fields_ = [("weights_path", C.c_char_p), ("model_w", C.c_int32), ("model_h", C.c_int32),
                ("conf_thr", C.c_float), ("iou_thr", C.c_float), ("max_batch", C.c_int32),
                ("max_dets", C.c_int32), ("device", C.c_int32), ("dtype", C.c_int32),
                ("warmup_runs", C.c_int32), ("use_graph", C.c_int32), ("flags", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("inference_count", C.c_uint64), ("inference_errors", C.c_uint64),
                ("total_preprocess_ms", C.c_double), ("total_forward_ms", C.c_double),
                ("total_postprocess_ms", C.c_double), ("last_detect_ms", C.c_double),
                ("sampled_frames", C.c_uint64), ("sampled_preprocess_ms", C.c_double), ("sampled_forward_ms", C.c_double),
                ("sampled_postprocess_ms", C.c_double), ("batches", C.c_uint64), ("graph_replays", C.c_uint64), ("eager_batches", C.c_uint64)]


class OpInfo(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("kind", C.c_int32), ("pad_", C.c_int32),
                ("flops_per_frame", C.c_double), ("bytes_per_frame", C.c_double)]


class LaunchInfo(C.Structure):
    _fields_ = [("covered_by", C.c_int32), ("n_ops", C.c_int32), ("flops_per_frame", C.c_double),
                ("bytes_unfused_per_frame", C.c_double), ("bytes_fused_per_frame", C.c_double), ("weight_bytes", C.c_double)]


class ZlyError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"zly error {code}: {msg}")
        self.code = code
        self.message = msg


_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Loads libzly.so and declares the prototypes.  Raises if the extension has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("ZLY_LIB") or LIB_PATH          # ZLY_LIB: A/B two builds on one box (dev aid)
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not found: build the HIP extension first (make, or __graft_entry__.build())")
    lib = C.CDLL(p)
    vp, i32, u32, f32, sz = C.c_void_p, C.c_int32, C.c_uint32, C.c_float, C.c_size_t
    pi32 = C.POINTER(C.c_int32)
    lib.zly_default_config.argtypes = [C.POINTER(Config)]; lib.zly_default_config.restype = None
    lib.zly_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]; lib.zly_create.restype = i32
    lib.zly_destroy.argtypes = [vp]; lib.zly_destroy.restype = i32
    lib.zly_last_error.argtypes = []; lib.zly_last_error.restype = C.c_char_p
    lib.zly_version.argtypes = []; lib.zly_version.restype = C.c_char_p
    lib.zly_detect.argtypes = [vp, vp, sz, i32, i32, vp, i32, pi32]; lib.zly_detect.restype = i32
    lib.zly_detect_batch.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz), pi32, pi32, vp, i32, pi32]; lib.zly_detect_batch.restype = i32
    lib.zly_submit.argtypes = [vp, vp, sz, i32, i32, C.POINTER(C.c_uint64)]; lib.zly_submit.restype = i32
    lib.zly_submit_try.argtypes = [vp, vp, sz, i32, i32, C.POINTER(C.c_uint64)]; lib.zly_submit_try.restype = i32
    lib.zly_poll.argtypes = [vp, C.c_uint64]; lib.zly_poll.restype = i32
    lib.zly_wait.argtypes = [vp, C.c_uint64, vp, i32, pi32]; lib.zly_wait.restype = i32
    lib.zly_detect_device.argtypes = [vp, i32, vp, i32, i32, vp, u32, vp]; lib.zly_detect_device.restype = i32
    lib.zly_slab_bytes.argtypes = [vp]; lib.zly_slab_bytes.restype = sz
    lib.zly_read_slabs.argtypes = [vp, i32, vp]; lib.zly_read_slabs.restype = i32
    lib.zly_sync.argtypes = [vp]; lib.zly_sync.restype = i32
    lib.zly_join.argtypes = [vp, vp, i32]; lib.zly_join.restype = i32
    lib.zly_preprocess.argtypes = [vp, vp, sz, i32, i32, vp]; lib.zly_preprocess.restype = i32
    lib.zly_forward.argtypes = [vp, i32, vp, vp]; lib.zly_forward.restype = i32
    lib.zly_head_tensor.argtypes = [vp, i32, vp]; lib.zly_head_tensor.restype = i32
    lib.zly_postprocess.argtypes = [vp, vp, i32, i32, i32, i32, f32, f32, vp, i32, pi32, pi32]; lib.zly_postprocess.restype = i32
    lib.zly_debug_tap.argtypes = [vp, C.c_char_p, i32, vp, sz, pi32, pi32, pi32]; lib.zly_debug_tap.restype = i32
    lib.zly_num_classes.argtypes = [vp]; lib.zly_num_classes.restype = i32
    lib.zly_num_anchors.argtypes = [vp]; lib.zly_num_anchors.restype = i32
    lib.zly_weights_fp8.argtypes = [vp]; lib.zly_weights_fp8.restype = i32
    lib.zly_num_ops.argtypes = [vp]; lib.zly_num_ops.restype = i32
    lib.zly_op_info_at.argtypes = [vp, i32, C.POINTER(OpInfo)]; lib.zly_op_info_at.restype = i32
    lib.zly_launch_info_at.argtypes = [vp, i32, i32, C.POINTER(LaunchInfo)]; lib.zly_launch_info_at.restype = i32
    lib.zly_op_kernel_name.argtypes = [vp, i32, i32, C.c_char_p, sz]; lib.zly_op_kernel_name.restype = i32
    lib.zly_profile_ops.argtypes = [vp, i32, vp, i32, i32, i32, vp]; lib.zly_profile_ops.restype = i32
    lib.zly_get_stats.argtypes = [vp, C.POINTER(Stats)]; lib.zly_get_stats.restype = i32
    if path is None:
        _lib = lib
    return lib


def _check(lib, rc: int):
    if rc != OK:
        raise ZlyError(rc, (lib.zly_last_error() or b"").decode(errors="replace"))


class Engine:
    """Thin object wrapper over a zly_engine handle."""

    def __init__(self, weights: Optional[str] = None, model_w: int = 416, model_h: int = 416, conf_thr: float = 0.5,
                 iou_thr: float = 0.45, max_batch: int = 1, max_dets: int = 64, device: int = 0,
                 dtype: int = DTYPE_BF16, warmup_runs: int = 1, use_graph: bool = True, flags: int = 0):
        self.lib = load_library()
        cfg = Config()
        self.lib.zly_default_config(C.byref(cfg))
        self._wpath = (weights or DEFAULT_WEIGHTS).encode()
        cfg.weights_path = self._wpath
        cfg.model_w, cfg.model_h = model_w, model_h
        cfg.conf_thr, cfg.iou_thr = conf_thr, iou_thr
        cfg.max_batch, cfg.max_dets, cfg.device, cfg.dtype = max_batch, max_dets, device, dtype
        cfg.warmup_runs, cfg.use_graph = warmup_runs, 1 if use_graph else 0
        cfg.flags = flags
        self.cfg = cfg
        h = C.c_void_p()
        _check(self.lib, self.lib.zly_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.model_w, self.model_h, self.max_batch, self.max_dets = model_w, model_h, max_batch, max_dets
        self.nc = self.lib.zly_num_classes(h)
        self.weights_fp8 = bool(self.lib.zly_weights_fp8(h))
        self.N = self.lib.zly_num_anchors(h)
        self.slab_bytes = int(self.lib.zly_slab_bytes(h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.zly_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- whole path --------------------------------------------------------------------------------
    def detect(self, frame: np.ndarray, cap: Optional[int] = None, nbytes: Optional[int] = None,
               w: Optional[int] = None, h: Optional[int] = None) -> Tuple[np.ndarray, int]:
        """frame: u8 [h][w][3] BGR.  -> (detections[min(n,cap)], n)."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        hh, ww = (frame.shape[0], frame.shape[1]) if frame.ndim == 3 else (h, w)
        ww = w if w is not None else ww
        hh = h if h is not None else hh
        cap = cap or self.max_dets
        out = np.zeros(cap, dtype=DET_DTYPE)
        n = C.c_int32(0)
        nb = frame.nbytes if nbytes is None else nbytes
        _check(self.lib, self.lib.zly_detect(self.h, frame.ctypes.data, nb, ww, hh, out.ctypes.data, cap, C.byref(n)))
        return out[:min(n.value, cap)], n.value

    def detect_batch(self, frames: Sequence[np.ndarray], cap: Optional[int] = None) -> List[Tuple[np.ndarray, int]]:
        frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        n = len(frames)
        cap = cap or self.max_dets
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        nbytes = (C.c_size_t * n)(*[f.nbytes for f in frames])
        ws = (C.c_int32 * n)(*[f.shape[1] for f in frames])
        hs = (C.c_int32 * n)(*[f.shape[0] for f in frames])
        out = np.zeros((n, cap), dtype=DET_DTYPE)
        n_out = (C.c_int32 * n)()
        _check(self.lib, self.lib.zly_detect_batch(self.h, n, ptrs, nbytes, ws, hs, out.ctypes.data, cap, n_out))
        return [(out[i, :min(n_out[i], cap)], int(n_out[i])) for i in range(n)]

    # -- asynchronous, pipelined host-to-host path ---------------------------------------------------
    def submit(self, frame: np.ndarray, nbytes: Optional[int] = None) -> int:
        """copies the frame into the engine's pinned staging ring (on this thread) and returns a ticket"""
        t = C.c_uint64(0)
        _check(self.lib, self.lib.zly_submit(self.h, frame.ctypes.data, frame.nbytes if nbytes is None else nbytes,
                                             frame.shape[1], frame.shape[0], C.byref(t)))
        return t.value

    def poll(self, ticket: int) -> bool:
        rc = self.lib.zly_poll(self.h, ticket)
        if rc not in (OK, PENDING):
            _check(self.lib, rc)
        return rc == OK

    def wait(self, ticket: int, cap: Optional[int] = None) -> Tuple[np.ndarray, int]:
        cap = cap or self.max_dets
        out = np.zeros(cap, dtype=DET_DTYPE)
        n = C.c_int32(0)
        _check(self.lib, self.lib.zly_wait(self.h, ticket, out.ctypes.data, cap, C.byref(n)))
        return out[:min(n.value, cap)], n.value

    def detect_device(self, d_frames_ptr: int, n: int, w: int, h: int, d_slabs_ptr: int = 0, tag0: int = 0, stream: int = 0):
        _check(self.lib, self.lib.zly_detect_device(self.h, n, d_frames_ptr, w, h, d_slabs_ptr or None, tag0, stream or None))

    def sync(self):
        _check(self.lib, self.lib.zly_sync(self.h))

    def join(self, stream: int = 0, lag: int = 0):
        """FLAG_ASYNC_NMS: make `stream` wait (on the device) for the NMS of every call so far but the last `lag`"""
        _check(self.lib, self.lib.zly_join(self.h, stream or None, lag))

    def read_slabs(self, n: int) -> List[Tuple[np.ndarray, np.ndarray]]:
        raw = np.zeros(n * self.slab_bytes, dtype=np.uint8)
        _check(self.lib, self.lib.zly_read_slabs(self.h, n, raw.ctypes.data))
        return parse_slabs(raw, n, self.max_dets)

    # -- stage level -------------------------------------------------------------------------------
    def preprocess(self, frame: np.ndarray, nbytes: Optional[int] = None, w: Optional[int] = None, h: Optional[int] = None) -> np.ndarray:
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        hh = h if h is not None else frame.shape[0]
        ww = w if w is not None else frame.shape[1]
        out = np.zeros((3, self.model_h, self.model_w), dtype=np.float32)
        nb = frame.nbytes if nbytes is None else nbytes
        _check(self.lib, self.lib.zly_preprocess(self.h, frame.ctypes.data, nb, ww, hh, out.ctypes.data))
        return out

    def forward(self, images_nchw: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(images_nchw, dtype=np.float32)
        n = x.shape[0]
        out = np.zeros((n, 4 + self.nc, self.N), dtype=np.float32)
        _check(self.lib, self.lib.zly_forward(self.h, n, x.ctypes.data, out.ctypes.data))
        return out

    def head_tensor(self, idx: int = 0) -> np.ndarray:
        out = np.zeros((4 + self.nc, self.N), dtype=np.float32)
        _check(self.lib, self.lib.zly_head_tensor(self.h, idx, out.ctypes.data))
        return out

    def postprocess(self, head: np.ndarray, img_w: int, img_h: int, conf_thr: float = 0.5, iou_thr: float = 0.45,
                    cap: Optional[int] = None) -> Tuple[np.ndarray, int, int]:
        head = np.ascontiguousarray(head, dtype=np.float32)
        ncls, nbox = head.shape[0] - 4, head.shape[1]
        cap = cap or max(nbox, 1)
        out = np.zeros(cap, dtype=DET_DTYPE)
        n, nc = C.c_int32(0), C.c_int32(0)
        _check(self.lib, self.lib.zly_postprocess(self.h, head.ctypes.data, ncls, nbox, img_w, img_h, conf_thr, iou_thr,
                                                  out.ctypes.data, cap, C.byref(n), C.byref(nc)))
        return out[:min(n.value, cap)], n.value, nc.value

    def tap(self, name: str, idx: int = 0) -> np.ndarray:
        cap = 4 * 1024 * 1024
        buf = np.zeros(cap, dtype=np.float32)
        c, h, w = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _check(self.lib, self.lib.zly_debug_tap(self.h, name.encode(), idx, buf.ctypes.data, cap, C.byref(c), C.byref(h), C.byref(w)))
        return buf[:c.value * h.value * w.value].reshape(c.value, h.value, w.value).copy()

    # -- introspection -----------------------------------------------------------------------------
    def ops(self) -> List[dict]:
        out = []
        for i in range(self.lib.zly_num_ops(self.h)):
            info = OpInfo()
            _check(self.lib, self.lib.zly_op_info_at(self.h, i, C.byref(info)))
            out.append(dict(name=info.name.decode(), kind=info.kind, flops=info.flops_per_frame, bytes=info.bytes_per_frame))
        return out

    def launches(self, n: int) -> List[dict]:
        """launch groups at batch size n: one record per op (covered_by != index: the op runs inside another op's launch)"""
        out = []
        for i in range(self.lib.zly_num_ops(self.h)):
            li = LaunchInfo()
            _check(self.lib, self.lib.zly_launch_info_at(self.h, i, n, C.byref(li)))
            out.append({k: getattr(li, k) for k, _ in LaunchInfo._fields_})
        return out

    def op_kernels(self, n: int) -> List[str]:
        """kernel (and tile shape) every op launches at batch size n"""
        out = []
        buf = C.create_string_buffer(96)
        for i in range(self.lib.zly_num_ops(self.h)):
            _check(self.lib, self.lib.zly_op_kernel_name(self.h, i, n, buf, 96))
            out.append(buf.value.decode())
        return out

    def profile_ops(self, d_frames_ptr: int, n: int, w: int, h: int, reps: int = 5) -> np.ndarray:
        ms = np.zeros(self.lib.zly_num_ops(self.h), dtype=np.float32)
        _check(self.lib, self.lib.zly_profile_ops(self.h, n, d_frames_ptr, w, h, reps, ms.ctypes.data))
        return ms

    def stats(self) -> dict:
        s = Stats()
        _check(self.lib, self.lib.zly_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}


def parse_slabs(raw: np.ndarray, n: int, cap: int) -> List[Tuple[np.ndarray, np.ndarray]]:
    """raw u8 [n * slab_bytes] -> [(header record, detections[min(n_kept, cap)])]."""
    sb = SLAB_HDR_DTYPE.itemsize + cap * DET_DTYPE.itemsize
    raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(n, sb)
    out = []
    for i in range(n):
        hdr = raw[i, :SLAB_HDR_DTYPE.itemsize].view(SLAB_HDR_DTYPE)[0]
        dets = raw[i, SLAB_HDR_DTYPE.itemsize:].view(DET_DTYPE)
        out.append((hdr, dets[:min(int(hdr["n_kept"]), cap)].copy()))
    return out
