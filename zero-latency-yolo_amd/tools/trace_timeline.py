#!/usr/bin/env python3
"""Timeline of ONE batch step from a rocprofv3 --kernel-trace CSV of bench.py: every dispatch's start/end relative to the
step's first kernel, in start order, with the number of kernels running at its start -- shows what overlaps what (side
streams, deferred NMS) and where the chip idles.   usage: trace_timeline.py kernel_trace.csv [step_index_from_end=3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
starts = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("preprocess_kernel", "stem_fused_kernel", "stem_model1_kernel"))]
i0, i1 = starts[-back - 1], starts[-back]
t0 = int(rows[i0]["Start_Timestamp"])
step = rows[i0:i1 + 3]
print(f"# step of {i1 - i0} dispatches, {(int(rows[i1]['Start_Timestamp']) - t0) / 1000:.1f} us to the next step's first kernel")
ends = []
dur_sum = gap_sum = 0.0
for r in step:
    st, en = (int(r["Start_Timestamp"]) - t0) / 1000, (int(r["End_Timestamp"]) - t0) / 1000
    live = sum(1 for e in ends if e > st)
    gap = st - max(ends) if ends else 0.0           # idle time between the end of everything launched before and this dispatch's begin (negative: overlap)
    ends.append(en)
    if len(ends) <= i1 - i0:
        dur_sum += en - st
        gap_sum += max(gap, 0.0)
    name = r["Kernel_Name"].replace("zly::", "").replace("void ", "")[:58]
    print(f"{st:8.1f} -> {en:8.1f}  ({en - st:6.2f} us, gap {gap:+6.2f})  +{live}  {name}")
print(f"# this step: sum of dispatch durations {dur_sum:.1f} us, sum of idle gaps between dispatches {gap_sum:.1f} us, {i1 - i0} dispatches "
      f"-> {dur_sum / (i1 - i0):.2f} us inside a dispatch, {gap_sum / (i1 - i0):.2f} us between two on average")
