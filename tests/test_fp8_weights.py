"""fp8 weight files (BASELINE configs[4]: "YOLOv8-s 640x640 fp8 weights"): OCP e4m3 codes + one power-of-two exponent per output
channel (tools/zly_model.py), dequantised once at load (csrc/weights.cpp).  CPU: the quantiser against torch.float8_e4m3fn,
file round trip, size, exactness in bf16 (the engine's MFMA operands).  GPU: the engine on such a file against the oracle on
the same dequantised weights."""
import os

import numpy as np
import pytest
import torch

import zly_model as zm


def test_quantiser_matches_torch_float8_e4m3fn():
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((16, 7, 3, 3)) * 10.0 ** rng.uniform(-5, 2, 16)[:, None, None, None]).astype(np.float32)
    w[3] = 0                                                    # an all-zero filter
    ex, code = zm.quantize_fp8(w)
    scaled = (w.reshape(16, -1) / (2.0 ** ex.astype(np.float64))[:, None]).astype(np.float32)
    want = torch.from_numpy(scaled).to(torch.float8_e4m3fn).view(torch.uint8).numpy().reshape(code.shape)
    assert np.array_equal(code, want)
    table = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
    ok = ~np.isnan(zm.E4M3)
    assert np.array_equal(table[ok], zm.E4M3[ok]) and np.isnan(table[~ok]).all() and (~ok).sum() == 2
    dq = zm.dequantize_fp8(ex, code)
    amax = np.abs(w).reshape(16, -1).max(1)
    assert np.all(np.abs(dq - w).reshape(16, -1).max(1) <= amax / 14 + 1e-30)          # half an e4m3 step at the top binade
    assert np.array_equal(torch.from_numpy(dq).to(torch.bfloat16).float().numpy(), dq)   # exact in bf16: 3 mantissa bits x 2^e
    assert np.all(np.abs(scaled).max(1)[amax > 0] > 224 - 1e-3)                           # the exponent uses the top binade


def test_fp8_file_round_trip_and_size(tmp_path):
    spec = zm.build_spec("s")
    w = zm.synth_weights(spec, seed=9)
    p8, p32 = str(tmp_path / "s_fp8.zlyw"), str(tmp_path / "s_f32.zlyw")
    zm.write_zlyw(p8, spec, w, fp8=True)
    zm.write_zlyw(p32, spec, w)
    assert os.path.getsize(p8) < 0.26 * os.path.getsize(p32)                   # 11.2 MB vs 44.6 MB (11.2 MB fp8 for s)
    meta, got = zm.read_zlyw(p8)
    assert meta["ch"] == spec.ch and len(got) == len(spec.convs)
    for c in spec.convs:
        ex, code = zm.quantize_fp8(w[c.name][0])
        assert np.array_equal(got[c.name][0], zm.dequantize_fp8(ex, code)) and np.array_equal(got[c.name][1], w[c.name][1])
