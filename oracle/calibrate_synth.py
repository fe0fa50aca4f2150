"""Offline helper (test infrastructure): LSUV-style calibration of the synthetic weights.
Runs the CPU oracle once over the seeded calibration frames, rescaling every conv so that its
pre-activation std hits the target of zly_model.py (SYNTH_ACT_STD for the SiLU convs, DFL_MU_STD bins
for the regressed distance of the Gaussian DFL head, CLS_LOGIT_STD for the class logits), and prints the
SYNTH_GAIN table that is pasted into zero-latency-yolo_amd/tools/zly_model.py, the class-logit quantiles
CLS_LOGIT_SHIFT is chosen from, and the bf16 noise floor of the result (bf16-rounding oracle vs fp32 oracle).
Usage: python oracle/calibrate_synth.py [scale]      (then: python oracle/calibrate_synth.py --check)"""
import os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "zero-latency-yolo_amd", "tools"))
sys.path.insert(0, here)
import zly_model as zm
import yolov8_ref


def target(name):
    if name.startswith("model.22.cv2.") and name.endswith(".2"):
        # rows are 2a * i * w_mu: std over the 16 bins of i * m, m ~ N(0, s^2), is sqrt(mean(i^2)) * s
        return 2.0 * zm.DFL_ALPHA * float(np.sqrt(np.mean(np.arange(16.0) ** 2))) * zm.DFL_MU_STD
    if name.startswith("model.22.cv3.") and name.endswith(".2"):
        return zm.CLS_LOGIT_STD
    return zm.SYNTH_ACT_STD


def run(scale, check):
    spec = zm.build_spec(scale)
    wts = zm.synth_weights(spec, gains=None if check else {})
    meta = dict(nc=spec.nc, reg_max=spec.reg_max, ch=spec.ch, n_c2f=spec.n_c2f, convs=spec.convs)
    net = yolov8_ref.YoloV8Ref(meta, wts, "fp32")
    if not check:
        net.calib_target = target
    frames = zm.synth_frames(4, 416, 416, rects=False)   # homogeneous noise frames: stable statistics
    x = torch.from_numpy(frames[..., ::-1].copy()).permute(0, 3, 1, 2).float() / 255.0
    out = net.forward(x)
    if not check:
        print("SYNTH_GAIN: Dict[str, float] = {")
        for c in spec.convs:
            print(f'    "{c.name}": {net.calib_scale[c.name]:.4f},')
        print("}")
    else:
        for name, t in net.taps.items():
            print(f"{name:24s} mean {t.mean():8.3f} std {t.std():8.3f} absmax {t.abs().max():9.3f}")
    sc = out[:, 4:].amax(1)
    print("# frac anchors >= 0.5 per frame:", [round(float((s >= 0.5).float().mean()), 4) for s in sc])
    lg = torch.cat([net.taps[f"model.22.cv3.{l}.2"].reshape(4, spec.nc, -1) for l in range(3)], 2)
    q = torch.quantile(lg.amax(1).flatten(), torch.tensor([0.5, 0.9, 0.98, 0.99, 0.995]))
    print("# class logit mean/std", round(lg.mean().item(), 3), round(lg.std().item(), 3),
          "max-over-class quantiles(50,90,98,99,99.5):", [round(v, 3) for v in q.tolist()])
    print("# box w/h mean", out[:, 2:4].mean().item(), "min", out[:, 2:4].min().item(), "max", out[:, 2:4].max().item())
    if check:
        # bf16 noise floor of this model: the bf16-rounding oracle vs the fp32 oracle, both on the CPU
        w2 = {k: (w.numpy(), b.numpy()) for k, (w, b) in net.w.items()}
        r16 = yolov8_ref.YoloV8Ref(meta, w2, "bf16")
        o16 = r16.forward(x)
        db, ds = (out[:, :4] - o16[:, :4]).numpy(), (out[:, 4:] - o16[:, 4:]).numpy()
        print("# bf16 noise floor: box rms %.3f px max %.3f px; score rms %.5f max %.4f" %
              (np.sqrt((db ** 2).mean()), np.abs(db).max(), np.sqrt((ds ** 2).mean()), np.abs(ds).max()))


run(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "n", "--check" in sys.argv)
