#!/bin/bash
# usage: ab_env_b1.sh ENVNAME v1 v2 ...   A/B of one environment variable on the batch-1 latency path (one engine, device-resident frames), 4 alternating rounds;
# the value "-" = variable unset
name=$1; shift
for r in 1 2 3 4; do for v in "$@"; do
  if [ "$v" = "-" ]; then setv="-u $name"; else setv="$name=$v"; fi
  env $setv python3 bench.py --batch 1 --engines 1 --steps 200 --warmup 20 --blocks 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json;d=json.loads(sys.stdin.readline());print('$name=$(basename $v) round $r  b1',d['value'],'frames/s',d['ms_per_step'],'ms')"
done; done
