import os, sys, numpy as np, torch
sys.path.insert(0,'zero-latency-yolo_amd'); sys.path.insert(0,'zero-latency-yolo_amd/tools')
import zly, zly_model as zm
eng = zly.Engine(max_batch=64, max_dets=64, warmup_runs=2)
fr = torch.from_numpy(zm.synth_frames(64,416,416,seed=3,rects=False)).cuda()
ops = eng.ops()
for nb in (1, 64):
    ms = eng.profile_ops(fr.data_ptr(), nb, 416, 416, reps=5)
    print(f"# batch {nb} inner={os.environ.get('ZLY_PROFILE_INNER')}: total {ms.sum()*1e3:.1f} us")
    for o, m in zip(ops, ms):
        print(f"{o['name']:46s} {m*1e3:8.2f} us")
