"""Layer-by-layer comparison of the HIP engine against the CPU oracle (run on the GPU box):
    python tests/gpu_debug.py [fp32|bf16] [n_frames]
Prints max-abs / relative error of every conv output, the head tensor and the detections."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("zero-latency-yolo_amd", "zero-latency-yolo_amd/tools", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import zly, zly_model as zm, yolov8_ref
from oracle_lib import Oracle, det_fields_equal

def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    wp = zly.DEFAULT_WEIGHTS
    ref = yolov8_ref.load(wp, mode)
    eng = zly.Engine(wp, dtype=zly.DTYPE_FP32 if mode == "fp32" else zly.DTYPE_BF16, max_batch=nf, max_dets=512, use_graph=False, flags=zly.FLAG_DUMP_LOGITS)
    orc = Oracle()
    frames = zm.synth_frames(nf, 416, 416, seed=5, rects=False)
    x = np.stack([orc.preprocess(f, 416, 416)[1] for f in frames])
    g = np.stack([eng.preprocess(f) for f in frames])
    print("preprocess bit-exact:", np.array_equal(x, g))
    want = ref.forward(torch.from_numpy(x)).numpy()
    got = eng.forward(x)
    worst = 0
    for name in ["images"] + [c.name for c in ref.spec.values()]:
        if name == "images":
            continue
        for i in range(nf):
            t = ref.taps[name][i].numpy()
            gt = eng.tap(name, i)
            if gt.shape != t.shape:
                print(f"{name:26s} SHAPE {gt.shape} vs {t.shape}"); continue
            d = np.abs(gt - t)
            rel = d.max() / (np.abs(t).max() + 1e-12)
            flag = "  <<<<" if rel > (1e-4 if mode == "fp32" else 3e-2) else ""
            if i == 0 or flag:
                print(f"{name:26s} f{i} max|d| {d.max():.3e} rel {rel:.3e} ref absmax {np.abs(t).max():.3f}{flag}")
    d = np.abs(got - want)
    print("head box rows max|d|", d[:, :4].max(), "score rows max|d|", d[:, 4:].max())
    for i in range(nf):
        dets, n = eng.detect(frames[i], cap=512)
        odet = orc.postprocess(want[i], 416, 416)
        own = orc.postprocess(eng.head_tensor(0), 416, 416)
        print(f"frame {i}: gpu n={n} oracle(ref head) n={len(odet)} oracle(gpu head) n={len(own)} "
              f"bitexact-vs-own-head={det_fields_equal(dets, own[:512])}")
main()
