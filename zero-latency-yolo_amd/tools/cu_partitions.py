"""Experiment: n engine instances on one GPU, each confined to 1/n of the chip's compute units (hipExtStreamCreateWithCUMask,
ZLY_CU_PART=i/n), fed alternate batch-64 steps on their own streams.  A step's ~35 launch-latency-bound small-map layers
leave most of the chip idle; with the chip cut into slices, the other slices' steps fill it.  Prints frames/s for 1, 2, 4."""
import os, sys, time
import numpy as np, torch
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.dirname(os.path.abspath(__file__))]
import zly, zly_model as zm

def run(n_eng, B=64, steps=200, warmup=20):
    flags = zly.FLAG_NO_HEAD_TENSOR | zly.FLAG_ASYNC_NMS
    engs = []
    for i in range(n_eng):
        if n_eng > 1:
            os.environ["ZLY_CU_PART"] = f"{i}/{n_eng}"
        else:
            os.environ.pop("ZLY_CU_PART", None)
        engs.append(zly.Engine(dtype=zly.DTYPE_BF16, max_batch=B, max_dets=64, warmup_runs=2, flags=flags))
    frames = torch.from_numpy(zm.synth_frames(4 * B, 416, 416, seed=1, rects=False)).cuda()
    sets = [frames[i * B:(i + 1) * B] for i in range(4)]
    slabs = [[torch.zeros(B * e.slab_bytes, dtype=torch.uint8, device="cuda") for _ in range(3)] for e in engs]
    torch.cuda.synchronize()
    def go(k0, n):
        for k in range(k0, k0 + n):
            i = k % n_eng
            engs[i].detect_device(sets[k % 4].data_ptr(), B, 416, 416, d_slabs_ptr=slabs[i][(k // n_eng) % 3].data_ptr(), tag0=k)   # the engine's own (masked) stream
        for e in engs:
            e.sync()
    go(0, warmup); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(warmup, steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for e in engs: e.close()
    return B * steps / dt

for n in (1, 2, 4, 1, 2, 4):
    print(n, "CU partitions:", round(run(n)), "frames/s", flush=True)
