"""CPU tests of the drop-in boundary: libzly.so loads, exports exactly what include/zly.h declares,
struct layouts match the reference's Detection, and the engine fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import zly

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, "include", "zly.h")).read()


def test_library_exports_every_declared_symbol():
    lib = zly.load_library()
    declared = set(re.findall(r"\b(zly_[a-z0-9_]+)\s*\(", _header()))
    declared -= {"zly_slab_bytes(e)"}
    assert declared == set(zly.SYMBOLS), declared ^ set(zly.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.zly_version().startswith(b"zly-hip")


def test_detection_layout_matches_reference_types_h():
    # reference src/common/types.h:16-26: box@0, confidence@16, class_id@20, track_id@24, timestamp@32, sizeof 40
    d = zly.DET_DTYPE
    assert d.itemsize == 40
    assert [d.fields[k][1] for k in ("x", "y", "w", "h", "confidence", "class_id", "track_id", "timestamp")] == [0, 4, 8, 12, 16, 20, 24, 32]
    assert zly.SLAB_HDR_DTYPE.itemsize == 16


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "src", "common", "types.h")), reason="build container only: the reference tree is not on the GPU box")
def test_detection_layout_against_the_reference_header_itself():
    """tests/cpp/abi_vs_reference_types.cpp includes the reference's own src/common/types.h (read where it lies, never copied) and
    static_asserts sizeof / alignof / offsetof of zero_latency::Detection against zly_det: the compiler, not a table of numbers."""
    import subprocess
    for tu in ("abi_vs_reference_types.cpp", "abi_vs_reference_result.cpp"):       # result.h: the return codes against zero_latency::ErrorCode
        src = os.path.join(ROOT, "tests", "cpp", tu)
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(REFERENCE, "src", "common"), "-I", os.path.join(ROOT, "include"), src],
                           capture_output=True, text=True)
        assert r.returncode == 0, (tu, r.stderr)
    # the plugin's compat header re-declares the same record (the reference's headers do not compile as a whole): same layout as zly_det
    compat = open(os.path.join(ROOT, "zero-latency-yolo_amd", "host", "hip_inference_engine.cpp")).read()
    assert "static_assert(sizeof(zly_det) == sizeof(Detection)" in compat


def test_error_codes_match_reference_result_h():
    h = _header()
    for name, val in (("ZLY_ERR_NOT_INITIALIZED", 3), ("ZLY_ERR_INFERENCE", 200), ("ZLY_ERR_MODEL_NOT_FOUND", 201),
                      ("ZLY_ERR_MODEL_LOAD", 202), ("ZLY_ERR_INVALID_INPUT", 203), ("ZLY_ERR_SYSTEM", 300)):
        assert re.search(rf"#define\s+{name}\s+{val}\b", h), name


def test_default_config_is_the_reference_config():
    lib = zly.load_library()
    cfg = zly.Config()
    lib.zly_default_config(C.byref(cfg))
    assert (cfg.model_w, cfg.model_h) == (416, 416)                    # configs/server.json:30-31
    assert abs(cfg.conf_thr - 0.5) < 1e-7 and abs(cfg.iou_thr - 0.45) < 1e-7   # configs/server.json:7-8
    assert cfg.warmup_runs == 3                                       # onnx_engine.cpp:919-954


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(zly.ZlyError) as ei:
        zly.Engine()
    assert ei.value.code == zly.ERR_SYSTEM and "no CPU fallback" in ei.value.message


def test_bad_arguments_are_rejected_before_touching_the_device():
    lib = zly.load_library()
    cfg = zly.Config()
    lib.zly_default_config(C.byref(cfg))
    cfg.model_w = 417
    h = C.c_void_p()
    assert lib.zly_create(C.byref(cfg), C.byref(h)) == zly.ERR_INVALID_ARGUMENT
    assert lib.zly_detect(None, None, 0, 0, 0, None, 0, None) == zly.ERR_NOT_INITIALIZED
    assert lib.zly_slab_bytes(None) == 0


def test_parse_slabs():
    cap = 4
    sb = 16 + cap * 40
    raw = np.zeros(2 * sb, dtype=np.uint8)
    hdr = raw[:16].view(zly.SLAB_HDR_DTYPE); hdr["n_kept"] = 2; hdr["frame_tag"] = 7
    raw[sb:sb + 16].view(zly.SLAB_HDR_DTYPE)["n_kept"] = 9            # overflowed slab: only cap stored
    out = zly.parse_slabs(raw, 2, cap)
    assert len(out[0][1]) == 2 and out[0][0]["frame_tag"] == 7 and len(out[1][1]) == cap
