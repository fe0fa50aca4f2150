"""CPU tests of the model description, the ZLYW weight file and the torch oracle's plumbing."""
import numpy as np
import torch

import zly_model as zm


def test_layer_table_reproduces_published_sizes():
    n = zm.build_spec("n")
    assert len(n.convs) == 63
    assert n.params() + 16 == 3151904                      # ultralytics "YOLOv8n summary (fused)": 3,151,904 params (16 = DFL conv)
    assert abs(2 * n.macs(640, 640) / 1e9 - 8.7) < 0.05    # published 8.7 GFLOPs @640
    assert abs(n.macs(416, 416) / 1e9 - 1.847) < 1e-3
    s = zm.build_spec("s")
    assert abs(s.params() / 1e6 - 11.157) < 1e-2 and abs(2 * s.macs(640, 640) / 1e9 - 28.6) < 0.05
    assert n.num_anchors(416, 416) == 3549 and n.num_anchors(640, 640) == 8400
    assert n.head_c2 == 64 and n.head_c3 == 80


def test_zlyw_roundtrip(tmp_path):
    spec = zm.build_spec("n")
    w = zm.synth_weights(spec)
    p = str(tmp_path / "m.zlyw")
    zm.write_zlyw(p, spec, w)
    meta, w2 = zm.read_zlyw(p)
    assert meta["nc"] == 80 and meta["ch"] == spec.ch and meta["n_c2f"] == spec.n_c2f
    assert [c.name for c in meta["convs"]] == [c.name for c in spec.convs]
    for c in spec.convs:
        assert np.array_equal(w[c.name][0], w2[c.name][0]) and np.array_equal(w[c.name][1], w2[c.name][1])


def test_synthetic_weights_are_deterministic():
    spec = zm.build_spec("n")
    a, b = zm.synth_weights(spec), zm.synth_weights(spec)
    assert all(np.array_equal(a[k][0], b[k][0]) for k in a)
    f1, f2 = zm.synth_frames(2, 64, 48, seed=9), zm.synth_frames(2, 64, 48, seed=9)
    assert np.array_equal(f1, f2) and f1.shape == (2, 48, 64, 3) and f1.dtype == np.uint8


def test_oracle_forward_contract(ref_fp32, ref_bf16):
    """[B,3,H,W] -> [B,4+nc,N] with the layout postProcess indexes (onnx_engine.cpp:767-796); scores in
    [0,1]; boxes in model pixels; batch invariance; bf16 emulation stays close to fp32."""
    x = torch.from_numpy(zm.synth_frames(2, 96, 64, seed=3, rects=False)[..., ::-1].copy()).permute(0, 3, 1, 2).float() / 255
    y = ref_fp32.forward(x)
    n = (64 // 8) * (96 // 8) + (64 // 16) * (96 // 16) + (64 // 32) * (96 // 32)
    assert y.shape == (2, 84, n)
    assert float(y[:, 4:].min()) >= 0 and float(y[:, 4:].max()) <= 1
    assert float(y[:, 2:4].min()) > 0
    y0 = ref_fp32.forward(x[:1])
    assert torch.allclose(y0, y[:1], atol=1e-4)
    yb = ref_bf16.forward(x)
    assert float((yb[:, 4:] - y[:, 4:]).abs().max()) < 0.2


def test_stem_pixel_scale_needs_no_divide():
    """kernels_stem.hip normalises with one multiply: bf16(u8 * (1/255)) must equal bf16(u8 / 255) (reference
    onnx_engine.cpp:693 + the engine's bf16 rounding) for every byte value; in fp32 the two differ for 126 of them."""
    import numpy as np
    import torch
    i = np.arange(256, dtype=np.float32)
    div = (i / np.float32(255.0)).astype(np.float32)
    mul = (i * np.float32(1.0 / 255.0)).astype(np.float32)
    assert (div != mul).sum() > 100
    assert torch.equal(torch.from_numpy(div).to(torch.bfloat16), torch.from_numpy(mul).to(torch.bfloat16))
