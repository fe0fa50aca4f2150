#!/usr/bin/env python3
"""Child process of tests/test_rccl_gather.py (GPU box): the multi-GPU exchange step of bench.py on the REAL backend --
torch.distributed "nccl" (= RCCL), world 1, process group created before any other GPU call of the process.

Sequence per step k, exactly what bench.run_steps queues:  zly_detect_device(step k, deferred NMS)  ->  zly_join(stream,
lag = 1) orders the stream behind NMS(k-1)  ->  all_gather_into_tensor(slabs of step k-1), async, beside step k.
Checks: the gathered bytes, put back into global frame order by shard.global_order, equal the slabs the same engine
produces for the same frames synchronously (zly_read_slabs), for every step; frame tags are the global frame ids."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("zero-latency-yolo_amd", "zero-latency-yolo_amd/tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import shard            # noqa: E402
import zly              # noqa: E402
import zly_model as zm  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    world, n, steps = dist.get_world_size(), 16, 5
    eng = zly.Engine(max_batch=n, max_dets=64, warmup_runs=1, flags=zly.FLAG_ASYNC_NMS | zly.FLAG_NO_HEAD_TENSOR)
    sb = eng.slab_bytes
    sets = [torch.from_numpy(zm.synth_frames(n, 416, 416, seed=60 + i, rects=False)).cuda() for i in range(3)]
    # reference slabs: same engine, synchronous reads
    want = []
    for k in range(steps):
        eng.detect_device(sets[k % 3].data_ptr(), n, 416, 416, tag0=k * n * world)
        want.append(eng.read_slabs(n))
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sp = stream.cuda_stream
    slabs = [torch.zeros(n * sb, dtype=torch.uint8, device="cuda") for _ in range(3)]
    gathered = [torch.zeros(world * n * sb, dtype=torch.uint8, device="cuda") for _ in range(steps)]
    torch.cuda.synchronize()
    works = []
    for k in range(steps):
        eng.detect_device(sets[k % 3].data_ptr(), n, 416, 416, d_slabs_ptr=slabs[k % 3].data_ptr(), tag0=k * n * world, stream=sp)
        if k > 0:
            eng.join(sp, 1)
            works.append(dist.all_gather_into_tensor(gathered[k - 1], slabs[(k - 1) % 3], async_op=True))
    eng.join(sp, 0)
    works.append(dist.all_gather_into_tensor(gathered[steps - 1], slabs[(steps - 1) % 3], async_op=True))
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    total = 0
    for k in range(steps):
        g = shard.global_order(gathered[k], world * n, world, sb).cpu().numpy()
        got = zly.parse_slabs(g.reshape(-1), world * n, eng.max_dets)
        for i in range(n):
            (gh, gd), (wh, wd) = got[i], want[k][i]
            assert int(gh["frame_tag"]) == k * n * world + i, (k, i, int(gh["frame_tag"]))
            assert int(gh["n_kept"]) == int(wh["n_kept"]) and int(gh["n_candidates"]) == int(wh["n_candidates"]), (k, i)
            assert gd.tobytes() == wd.tobytes(), (k, i)
            total += int(gh["n_kept"])
    assert total > 0
    dist.barrier()
    eng.close()
    dist.destroy_process_group()
    print(f"RCCL gather ok: {steps} steps x {n} frames, {total} detections, backend {dist.Backend.NCCL}, world {world}")


if __name__ == "__main__":
    main()
