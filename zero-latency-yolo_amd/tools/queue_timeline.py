#!/usr/bin/env python3
"""Timeline slice of a rocprofv3 kernel trace with the hardware queue of every dispatch: start/end relative to the slice, queue id,
kernel; and per queue the busy fraction and mean gap between consecutive kernels.  usage: queue_timeline.py trace.csv [n_rows=140]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 140
sl = rows[-n - 300:-300]
t0 = int(sl[0]["Start_Timestamp"])
qk = "Queue_Id" if "Queue_Id" in sl[0] else "Queue_ID"
for r in sl:
    st, en = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{st:9.1f} -> {en:9.1f} ({en - st:6.1f})  q{r[qk]:>3s}  {r['Kernel_Name'].replace('void zly::','').replace('zly::','')[:50]}")
per = collections.defaultdict(list)
for r in rows[len(rows) // 2:]:
    per[r[qk]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for q, v in per.items():
    busy = sum(e - s for s, e in v); span = v[-1][1] - v[0][0]
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    print(f"queue {q}: {len(v)} kernels, busy {busy / span:.2f}, mean gap {sum(gaps) / max(1, len(gaps)) / 1e3:.2f} us, median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us")
