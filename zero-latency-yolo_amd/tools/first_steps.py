#!/usr/bin/env python3
"""Diagnostic: how long do the FIRST steps after engine creation take?  (round-3 question: why did the driver's
`bench.py --steps 20 --warmup 5` measure 1.24 ms/step when 200-step runs measure 0.69.)

Creates the bench's engines, then times consecutive blocks of `--block` batch-64 steps (sync on both sides of every block) and, per step,
the host time of the enqueue call.  Prints one line per block.  --idle S sleeps S seconds between two series (does an idle gap
bring the slow start back?)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ("zero-latency-yolo_amd", "zero-latency-yolo_amd/tools"):
    sys.path.insert(0, os.path.join(ROOT, _p))
import zly            # noqa: E402
import zly_model as zm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engines", type=int, default=3)
    ap.add_argument("--block", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=16)
    ap.add_argument("--idle", type=float, default=2.0)
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    B = a.batch
    t_c = time.perf_counter()
    flags = zly.FLAG_NO_HEAD_TENSOR | (zly.FLAG_SINGLE_CHAIN if a.engines > 1 else zly.FLAG_ASYNC_NMS)
    engs = [zly.Engine(None, dtype=zly.DTYPE_BF16, max_batch=B, max_dets=64, warmup_runs=3, flags=flags) for _ in range(a.engines)]
    print(f"create {a.engines} engines: {time.perf_counter() - t_c:.2f} s", flush=True)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sp = stream.cuda_stream
    frames = zm.synth_frames(4 * B, 416, 416, seed=20250328, rects=False)
    d_all = torch.from_numpy(frames).cuda()
    sets = [d_all[i * B:(i + 1) * B] for i in range(4)]
    sb = engs[0].slab_bytes
    slabs = [torch.zeros(B * sb, dtype=torch.uint8, device="cuda") for _ in range(2 * a.engines)]
    k = 0
    for series in range(2):
        print(f"-- series {series} ({'cold' if series == 0 else f'after {a.idle} s idle'})", flush=True)
        for b in range(a.blocks):
            torch.cuda.synchronize()
            host = []
            t0 = time.perf_counter()
            for _ in range(a.block):
                h0 = time.perf_counter()
                engs[k % a.engines].detect_device(sets[k % 4].data_ptr(), B, 416, 416, d_slabs_ptr=slabs[k % len(slabs)].data_ptr(), tag0=k * B,
                                                  stream=sp if a.engines == 1 else 0)
                host.append((time.perf_counter() - h0) * 1e3)
                k += 1
            for e in engs:
                e.join(sp)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            print(f"block {b:2d}: {dt / a.block:7.3f} ms/step   host enqueue per step: " + " ".join(f"{h:6.3f}" for h in host), flush=True)
        time.sleep(a.idle)
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
