"""Real-weights loader: tools/convert_weights.py + tools/onnx_min.py.
BN folding is checked against torch's own conv2d + batch_norm; the variant / class-count inference and both container
formats (state dict, ONNX) against models built here with the ultralytics module names.  No real export is available in
this image (no ultralytics, no onnx, the reference ships no model): parity with a real file is unpinned."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import convert_weights as cw
import onnx_min
import onnx_writer as ow
import zly_model as zm


def _unfused_state_dict(spec, seed):
    """ultralytics-style: Conv modules = .conv.weight + .bn.{weight,bias,running_mean,running_var,num_batches_tracked};
    the last conv of each Detect branch is a plain Conv2d with .weight/.bias; plus the DFL projection"""
    rng = np.random.default_rng(seed)
    sd = {}
    for c in spec.convs:
        w = (rng.standard_normal((c.cout, c.cin, c.k, c.k)) / np.sqrt(c.cin * c.k * c.k)).astype(np.float32)
        if c.act:
            sd[c.name + ".conv.weight"] = w
            sd[c.name + ".bn.weight"] = rng.uniform(0.5, 1.5, c.cout).astype(np.float32)
            sd[c.name + ".bn.bias"] = rng.standard_normal(c.cout).astype(np.float32) * 0.1
            sd[c.name + ".bn.running_mean"] = rng.standard_normal(c.cout).astype(np.float32) * 0.1
            sd[c.name + ".bn.running_var"] = rng.uniform(0.5, 2.0, c.cout).astype(np.float32)
            sd[c.name + ".bn.num_batches_tracked"] = np.int64(1000)
        else:
            sd[c.name + ".weight"] = w
            sd[c.name + ".bias"] = rng.standard_normal(c.cout).astype(np.float32)
    sd["model.22.dfl.conv.weight"] = np.arange(spec.reg_max, dtype=np.float32).reshape(1, spec.reg_max, 1, 1)
    return sd


@pytest.mark.parametrize("scale,nc", [("n", 80), ("n", 4), ("s", 80)])
def test_state_dict_bn_folding_matches_torch(scale, nc):
    spec = zm.build_spec(scale, nc)
    sd = _unfused_state_dict(spec, seed=3)
    got_spec, w = cw.from_state_dict(sd)
    assert (got_spec.scale, got_spec.nc, got_spec.reg_max) == (scale, nc, 16) and got_spec.convs == spec.convs
    rng = np.random.default_rng(4)
    for c in spec.convs[:12] + spec.convs[-6:]:
        x = torch.from_numpy(rng.standard_normal((1, c.cin, 9, 9)).astype(np.float32))
        wf, bf = (torch.from_numpy(a) for a in w[c.name])
        got = F.conv2d(x, wf, bf, stride=c.stride, padding=c.k // 2)
        if c.act:
            t = {k: torch.from_numpy(np.asarray(sd[c.name + ".bn." + k])) for k in ("weight", "bias", "running_mean", "running_var")}
            want = F.batch_norm(F.conv2d(x, torch.from_numpy(sd[c.name + ".conv.weight"]), None, stride=c.stride, padding=c.k // 2),
                                t["running_mean"], t["running_var"], t["weight"], t["bias"], training=False, eps=1e-3)
        else:
            want = F.conv2d(x, torch.from_numpy(sd[c.name + ".weight"]), torch.from_numpy(sd[c.name + ".bias"]))
        assert torch.allclose(got, want, rtol=1e-5, atol=2e-6), c.name


def test_rejects_wrong_models():
    spec = zm.build_spec("n")
    sd = _unfused_state_dict(spec, seed=5)
    bad = dict(sd); bad["model.22.dfl.conv.weight"] = np.ones((1, 16, 1, 1), np.float32)
    with pytest.raises(ValueError, match="arange"):
        cw.from_state_dict(bad)
    bad = dict(sd); bad["model.4.cv1.conv.weight"] = np.zeros((64, 65, 1, 1), np.float32)
    with pytest.raises(ValueError, match="model.4.cv1"):
        cw.from_state_dict(bad)
    bad = dict(sd); del bad["model.9.cv2.conv.weight"]
    with pytest.raises(KeyError):
        cw.from_state_dict(bad)
    bad = dict(sd); bad["model.0.conv.weight"] = np.zeros((24, 3, 3, 3), np.float32)
    with pytest.raises(ValueError, match="not a YOLOv8"):
        cw.from_state_dict(bad)


def _fused_onnx(tmp_path, spec, weights, named, styles=("raw",)):
    """graph in module order; Detect per level: box branch then class branch (Detect.forward)"""
    order = [c.name for c in spec.convs if not c.name.startswith("model.22.")] + \
        [f"model.22.cv{b}.{l}.{i}" for l in range(3) for b in (2, 3) for i in range(3)]
    act = {c.name: c.act for c in spec.convs}
    nodes, inits = [], []
    for k, name in enumerate(order):
        w, b = weights[name]
        if named:
            wn = name + (".conv.weight" if act[name] else ".weight")
            bn = wn[:-6] + "bias"
        else:
            wn, bn = f"onnx::Conv_{1000 + 2 * k}", f"onnx::Conv_{1001 + 2 * k}"
        inits.append(ow.tensor(wn, w, styles[k % len(styles)]))
        inits.append(ow.tensor(bn, b, "raw"))
        nodes.append(ow.node("Conv", [f"t{k}", wn, bn], [f"t{k + 1}"], f"/{name}/Conv"))
        if act[name]:
            nodes.append(ow.node("Sigmoid", [f"t{k + 1}"], [f"s{k}"]))
            nodes.append(ow.node("Mul", [f"t{k + 1}", f"s{k}"], [f"m{k}"]))
    dfl = np.arange(spec.reg_max, dtype=np.float32).reshape(1, spec.reg_max, 1, 1)
    inits.append(ow.tensor("model.22.dfl.conv.weight", dfl))
    nodes.append(ow.node("Conv", ["dfl_in", "model.22.dfl.conv.weight"], ["dfl_out"], "/model.22/dfl/conv/Conv"))
    inits.append(ow.tensor("/model.22/Constant_shape", np.array([1, 4, 16, -1], dtype=np.int64), "int64_data"))
    p = tmp_path / ("named.onnx" if named else "anon.onnx")
    p.write_bytes(ow.model(nodes, inits))
    return str(p)


@pytest.mark.parametrize("named", [True, False])
def test_onnx_to_zlyw_round_trip(tmp_path, named):
    spec = zm.build_spec("n", 4)                                   # the reference's CS 1.6 head: 4 classes -> [1, 8, N]
    weights = zm.synth_weights(spec, seed=21)
    path = _fused_onnx(tmp_path, spec, weights, named, styles=("raw", "float_data", "packed_dims"))
    inits, nodes = onnx_min.read_onnx(path)
    assert sum(n["op_type"] == "Conv" for n in nodes) == len(spec.convs) + 1
    assert inits["/model.22/Constant_shape"].tolist() == [1, 4, 16, -1]
    got_spec, got = cw.from_onnx(path)
    assert got_spec.convs == spec.convs and got_spec.nc == 4
    for c in spec.convs:
        assert np.array_equal(got[c.name][0], weights[c.name][0]) and np.array_equal(got[c.name][1], weights[c.name][1]), c.name
    out = str(tmp_path / "m.zlyw")
    assert cw.main(["--onnx", path, "--out", out]) == 0
    meta, back = zm.read_zlyw(out)
    assert meta["nc"] == 4 and all(np.array_equal(back[c.name][0], weights[c.name][0]) for c in spec.convs)


def test_state_dict_cli_and_oracle_forward(tmp_path):
    import yolov8_ref
    spec = zm.build_spec("n", 4)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in _unfused_state_dict(spec, seed=8).items()}
    p = str(tmp_path / "sd.pt")
    torch.save(sd, p)
    out = str(tmp_path / "m.zlyw")
    assert cw.main(["--state-dict", p, "--out", out]) == 0
    ref = yolov8_ref.load(out, "fp32")
    y = ref.forward(torch.rand(1, 3, 64, 64))
    assert tuple(y.shape) == (1, 8, 84) and torch.isfinite(y).all()         # [1, 4+nc, N]: 8x8 + 4x4 + 2x2 anchors


def test_onnx_reader_rejects_garbage(tmp_path):
    p = tmp_path / "x.onnx"
    p.write_bytes(b"\x08\x08")                                              # ir_version only, no graph
    with pytest.raises(ValueError, match="GraphProto"):
        onnx_min.read_onnx(str(p))
    p.write_bytes(ow._ld(7, ow._ld(5, ow._vi(1, 4) + ow._vi(2, 1) + ow._ld(9, b"\0" * 8) + ow._ld(8, b"w"))))   # 2 floats for dims [4]
    with pytest.raises(ValueError, match="elements"):
        onnx_min.read_onnx(str(p))


@pytest.mark.gpu
@pytest.mark.parametrize("named", [True, False])
def test_converted_onnx_runs_on_the_engine(tmp_path, named):
    """SURVEY 8f rank 3 end to end on the GPU: an ONNX file laid out like `yolo export format=onnx` writes it (reference
    start.sh:122-132: BN fused, Conv + Sigmoid + Mul nodes, initializers named by module path or anonymised) ->
    tools/convert_weights.py -> .zlyw -> libzly.so.  The HIP engine's head tensor must match the CPU oracle evaluated on the
    ORIGINAL tensors (the ones that went into the ONNX file, never through the converter): fp32 engine to SURVEY 8c's fp32
    tolerance, bf16 engine to the bf16 tolerance, and detect() must equal the oracle's post-processing of the engine's head."""
    import yolov8_ref
    import zly
    from oracle_lib import Oracle, det_fields_equal
    spec = zm.build_spec("n", 80)
    weights = zm.synth_weights(spec, seed=33)                       # calibrated synthetic weights (the bf16 tolerance presumes them), a seed of their own
    path = _fused_onnx(tmp_path, spec, weights, named, styles=("raw", "float_data", "packed_dims"))
    out = str(tmp_path / "converted.zlyw")
    assert cw.main(["--onnx", path, "--out", out]) == 0
    meta = dict(nc=spec.nc, reg_max=spec.reg_max, ch=spec.ch, n_c2f=spec.n_c2f, convs=spec.convs)
    ref = yolov8_ref.YoloV8Ref(meta, weights, "fp32")              # the original tensors
    oracle = Oracle()
    frames = zm.synth_frames(2, 416, 416, seed=44, rects=False)
    x = np.stack([oracle.preprocess(f, 416, 416)[1] for f in frames])
    want = ref.forward(torch.from_numpy(x)).numpy()
    e32 = zly.Engine(out, dtype=zly.DTYPE_FP32, max_batch=2, max_dets=256, warmup_runs=0)
    got = e32.forward(x)
    assert np.abs(got[:, :4] - want[:, :4]).max() <= 1e-3 and np.abs(got[:, 4:] - want[:, 4:]).max() <= 1e-4
    e32.close()
    e16 = zly.Engine(out, dtype=zly.DTYPE_BF16, max_batch=2, max_dets=256, warmup_runs=1)
    for i, f in enumerate(frames):
        dets, n = e16.detect(f, cap=256)
        head = e16.head_tensor(0)
        assert np.abs(head[:4] - want[i, :4]).max() <= 1.5 and np.abs(head[4:] - want[i, 4:]).max() <= 2e-2
        own = oracle.postprocess(head, 416, 416)
        assert n == len(own) and det_fields_equal(dets, own[:256])
    e16.close()
