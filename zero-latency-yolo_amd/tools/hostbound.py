"""Experiment: is the batch-64 step loop host-bound?  Host time to enqueue a step against the device time per step, one and three engines."""
import os, sys, time
import numpy as np, torch
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.dirname(os.path.abspath(__file__))]
import zly, zly_model as zm
B=64
for n_eng in (1,3):
    flags = zly.FLAG_NO_HEAD_TENSOR | (zly.FLAG_SINGLE_CHAIN if n_eng>1 else zly.FLAG_ASYNC_NMS)
    engs=[zly.Engine(dtype=zly.DTYPE_BF16, max_batch=B, max_dets=64, warmup_runs=2, flags=flags) for _ in range(n_eng)]
    frames = torch.from_numpy(zm.synth_frames(4*B,416,416,seed=1,rects=False)).cuda()
    sets=[frames[i*B:(i+1)*B] for i in range(4)]
    slabs=[torch.zeros(B*engs[0].slab_bytes,dtype=torch.uint8,device="cuda") for _ in range(8)]
    def go(steps):
        for k in range(steps):
            engs[k%n_eng].detect_device(sets[k%4].data_ptr(), B, 416, 416, d_slabs_ptr=slabs[k%8].data_ptr(), tag0=k)
    go(30); torch.cuda.synchronize()
    t0=time.perf_counter(); go(300); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(n_eng, "engines: host enqueue %.3f ms/step, total %.3f ms/step" % ((t1-t0)/300*1e3, (t2-t0)/300*1e3))
    for e in engs: e.close()
