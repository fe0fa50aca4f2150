#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py per position in the step.
usage: trace_summary.py kernel_trace.csv kernels_per_step [out.txt]
The last 60 % of the dispatches are split into steps of `kernels_per_step` dispatches (the graph replays
them in a fixed order); per position: kernel name, grid, mean duration, and the mean gap to the previous
kernel's end.  Durations under ~4.5 us are at the profiler's floor."""
import csv, sys, collections
path, kps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# align on the preprocess kernel (first kernel of a step)
starts = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("preprocess_kernel", "stem_fused_kernel", "stem_model1_kernel"))]
starts = [i for i in starts if i + kps <= len(rows)]
starts = starts[len(starts) // 3:]
pos = collections.defaultdict(lambda: [0, 0.0, 0.0, "", ""])
nsteps = 0
for s in starts:
    step = rows[s:s + kps]
    if sum((any(k in r["Kernel_Name"] for k in ("preprocess_kernel", "stem_fused_kernel", "stem_model1_kernel"))) for r in step) != 1:
        continue
    nsteps += 1
    prev_end = None
    for j, r in enumerate(step):
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        p = pos[j]
        p[0] += 1; p[1] += en - st
        if prev_end is not None: p[2] += st - prev_end
        p[3] = r["Kernel_Name"]; p[4] = f'{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}/{r["Workgroup_Size_X"]}'
        prev_end = en
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
tot = gap = 0
out.write(f"# {path}: {nsteps} steps of {kps} dispatches\n")
for j in range(kps):
    n, d, g, name, grid = pos[j]
    if not n: continue
    short = name.replace("void zly::", "").replace("zly::", "")[:58]
    out.write(f"{j:3d} {short:58s} {grid:>22s} {d / n / 1e3:8.2f} us  gap {g / n / 1e3:6.2f} us\n")
    tot += d / n; gap += g / n
out.write(f"sum of kernel durations {tot / 1e3:.1f} us + gaps {gap / 1e3:.1f} us = {(tot + gap) / 1e3:.1f} us per step\n")
