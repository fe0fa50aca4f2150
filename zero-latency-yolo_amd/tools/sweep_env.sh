#!/bin/bash
# Dev aid: bench.py (batch 64, no extras) under a list of environment settings; one line per setting in gpurun_out/sweep.txt.
# usage: bash zero-latency-yolo_amd/tools/sweep_env.sh "VAR=1" "A=2 B=3" ...
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg" >> gpurun_out/sweep.txt
  env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/sweep.txt
done
