#!/bin/bash
# usage: ab.sh name ENV=.. ENV=..   -> runs bench (no extras except roofline) and prints value + selected rows
name=$1; shift
env ZLY_BENCH_NO_H2H=1 "$@" timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err
python - <<PY
import json
j=json.load(open("gpurun_out/ab/$name.json"))
r=j["roofline"]
print("$name", j["value"], j["ms_per_step"], "conv_ms", r["kernel_ms_per_step"], "b1", j["latency_path_b1"]["value"])
PY
