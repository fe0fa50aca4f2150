"""Experiment: k engine instances on one GPU (ZLY_FLAG_SINGLE_CHAIN: one stream + the NMS stream each), fed alternate batch-64 steps.
Run with GPU_MAX_HW_QUEUES=8 so that every stream owns a hardware queue.  usage: multi_engine.py [async_nms=1] [reps=4]"""
import os, sys, time
import numpy as np, torch
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.dirname(os.path.abspath(__file__))]
import zly, zly_model as zm

def run(n_eng, B=64, steps=200, warmup=20, async_nms=True, own_stream=True):
    flags = zly.FLAG_NO_HEAD_TENSOR | zly.FLAG_SINGLE_CHAIN | (zly.FLAG_ASYNC_NMS if async_nms else 0)
    engs = [zly.Engine(dtype=zly.DTYPE_BF16, max_batch=B, max_dets=64, warmup_runs=2, flags=flags) for _ in range(n_eng)]
    frames = torch.from_numpy(zm.synth_frames(4 * B, 416, 416, seed=1, rects=False)).cuda()
    sets = [frames[i * B:(i + 1) * B] for i in range(4)]
    slabs = [[torch.zeros(B * e.slab_bytes, dtype=torch.uint8, device="cuda") for _ in range(3)] for e in engs]
    torch.cuda.synchronize()
    def go(k0, n):
        for k in range(k0, k0 + n):
            i = k % n_eng
            engs[i].detect_device(sets[k % 4].data_ptr(), B, 416, 416, d_slabs_ptr=slabs[i][(k // n_eng) % 3].data_ptr(), tag0=k)   # the engine's own stream
        for e in engs:
            e.sync()
    go(0, warmup); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(warmup, steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for e in engs: e.close()
    return B * steps / dt

a = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    print("rep", rep, {n: round(run(n, async_nms=bool(a))) for n in [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "1,2,3".split(","))]}, flush=True)
