"""ctypes wrapper over oracle/_build/libzly_oracle.so (TEST INFRASTRUCTURE: the CPU oracle).
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "_build", "libzly_oracle.so")

DET_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("w", "<f4"), ("h", "<f4"), ("confidence", "<f4"),
                      ("class_id", "<i4"), ("track_id", "<u4"), ("pad_", "<u4"), ("timestamp", "<u8")])


class Oracle:
    def __init__(self, path=LIB_PATH):
        self.lib = lib = C.CDLL(path)
        vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        pi = C.POINTER(C.c_int)
        lib.zlyo_preprocess.argtypes = [vp, sz, i32, i32, i32, i32, vp]; lib.zlyo_preprocess.restype = i32
        lib.zlyo_iou.argtypes = [vp, vp]; lib.zlyo_iou.restype = f32
        lib.zlyo_decode.argtypes = [vp, i32, i32, i32, i32, f32, vp, i32, pi]; lib.zlyo_decode.restype = i32
        lib.zlyo_nms.argtypes = [vp, i32, f32, vp, pi]; lib.zlyo_nms.restype = i32
        lib.zlyo_postprocess.argtypes = [vp, i32, i32, i32, i32, f32, f32, vp, pi]; lib.zlyo_postprocess.restype = i32
        lib.zlyo_sizeof_det.argtypes = []; lib.zlyo_sizeof_det.restype = sz
        assert lib.zlyo_sizeof_det() == DET_DTYPE.itemsize == 40

    def preprocess(self, frame, tw, th, nbytes=None, w=None, h=None):
        """-> (rc, fp32 [3][th][tw])"""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        hh = h if h is not None else frame.shape[0]
        ww = w if w is not None else frame.shape[1]
        out = np.zeros((3, th, tw), dtype=np.float32)
        rc = self.lib.zlyo_preprocess(frame.ctypes.data, frame.nbytes if nbytes is None else nbytes, ww, hh, tw, th, out.ctypes.data)
        return rc, out

    def iou(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        return float(self.lib.zlyo_iou(a.ctypes.data, b.ctypes.data))

    def decode(self, head, img_w, img_h, conf_thr=0.5):
        head = np.ascontiguousarray(head, dtype=np.float32)
        nc, nb = head.shape[0] - 4, head.shape[1]
        out = np.zeros(max(nb, 1), dtype=DET_DTYPE)
        n = C.c_int(0)
        self.lib.zlyo_decode(head.ctypes.data, nc, nb, img_w, img_h, conf_thr, out.ctypes.data, nb, C.byref(n))
        return out[:n.value].copy()

    def nms(self, dets, iou_thr=0.45):
        dets = np.ascontiguousarray(dets, dtype=DET_DTYPE).copy()
        out = np.zeros(max(len(dets), 1), dtype=DET_DTYPE)
        n = C.c_int(0)
        self.lib.zlyo_nms(dets.ctypes.data, len(dets), iou_thr, out.ctypes.data, C.byref(n))
        return out[:n.value].copy()

    def postprocess(self, head, img_w, img_h, conf_thr=0.5, iou_thr=0.45):
        head = np.ascontiguousarray(head, dtype=np.float32)
        nc, nb = head.shape[0] - 4, head.shape[1]
        out = np.zeros(max(nb, 1), dtype=DET_DTYPE)
        n = C.c_int(0)
        self.lib.zlyo_postprocess(head.ctypes.data, nc, nb, img_w, img_h, conf_thr, iou_thr, out.ctypes.data, C.byref(n))
        return out[:n.value].copy()


def det_fields_equal(a, b):
    """bit-exact comparison of everything but the wall-clock timestamp"""
    if len(a) != len(b):
        return False
    for k in ("x", "y", "w", "h", "confidence"):
        if not np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)):
            return False
    return np.array_equal(a["class_id"], b["class_id"]) and np.array_equal(a["track_id"], b["track_id"])
