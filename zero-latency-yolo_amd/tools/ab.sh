#!/bin/bash
# usage: ab.sh name [ENV=.. ...]   -> the headline configuration (three engines, batch 64, median of 12 blocks of 20 steps) with the given
# environment, then the per-launch table of one engine; prints ms/step, conv sum and the rows whose op name matches $AB_ROWS (regex)
name=$1; shift
mkdir -p gpurun_out/ab
env ZLY_BENCH_NO_H2H=1 "$@" timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --blocks 12 --no-cpu-baseline --per-launch-out gpurun_out/ab/$name.pl.json > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err
python3 - <<PY
import json, re, os
j=json.load(open("gpurun_out/ab/$name.json"))
r=j["roofline"]
print("$name", j["value"], "fps", j["ms_per_step"], "ms/step  conv_ms", r["kernel_ms_per_step"], "launches", r["launches_per_step"], " b1", j["latency_path_b1"]["value"], "fps")
rows=os.environ.get("AB_ROWS")
if rows:
    pl=json.load(open("gpurun_out/ab/$name.pl.json"))
    for x in pl["per_launch"]:
        if re.search(rows, x["op"]): print("   ", x["op"][:40].ljust(40), x["kernel"][:50].ljust(50), x["us"])
PY
