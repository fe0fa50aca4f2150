#!/bin/bash
# usage: sweep_batch.sh [ENV=.. ...] -> frames/s of one engine at batch 2 .. 64 with and without the given environment (same box)
run() { b=$1; shift; env ZLY_BENCH_NO_H2H=1 "$@" timeout -k 10 200 python3 bench.py --batch $b --engines 1 --steps 50 --warmup 10 --blocks 6 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for b in 2 4 8 16 32 64; do
  echo "batch $b  base: $(run $b A=1)   with $*: $(run $b "$@")"
done
