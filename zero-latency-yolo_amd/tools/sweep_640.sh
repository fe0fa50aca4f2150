#!/bin/bash
# usage: sweep_640.sh [n|s] ENV=.. [ENV=.. ...] -> frames/s of the 640 x 640 batch-32 configuration (three engines) once per setting, baselines in between
scale=${1:-n}; shift
run() { env ZLY_BENCH_NO_H2H=1 "$@" timeout -k 10 300 python3 bench.py --size 640 --batch 32 --scale $scale --steps 20 --warmup 5 --blocks 10 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
echo "base: $(run A=1)"
for e in "$@"; do echo "$e: $(run $e)"; done
echo "base: $(run A=1)"
