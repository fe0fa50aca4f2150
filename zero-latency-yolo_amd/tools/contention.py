#!/usr/bin/env python3
"""Per-kernel mean duration in two rocprofv3 kernel traces of bench.py (e.g. --engines 1 vs --engines 3): which kernels stretch when
the chains of several engines overlap.  usage: contention.py trace_a.csv trace_b.csv"""
import csv, sys, collections
def load(p):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) < 64 * 256: continue
        d[r["Kernel_Name"].replace("void zly::", "").replace("zly::", "")[:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
tot_a = tot_b = 0
rows = []
for k in a:
    if k not in b: continue
    ma, mb = sum(a[k]) / len(a[k]) / 1e3, sum(b[k]) / len(b[k]) / 1e3
    per_step_a = sum(a[k]) / 1e3; per_step_b = sum(b[k]) / 1e3
    rows.append((mb * len(b[k]), k, len(a[k]), ma, len(b[k]), mb))
for _, k, na, ma, nb, mb in sorted(rows, reverse=True):
    print(f"{k:60s} n={na:5d} {ma:8.2f} us   n={nb:5d} {mb:8.2f} us   x{mb / ma:5.2f}")
