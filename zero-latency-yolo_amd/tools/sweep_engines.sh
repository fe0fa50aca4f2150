#!/bin/bash
# Round 3 result (one box, frames/s): 2 engines 96.5k, 3: 100.5-100.8k, 4: 91.5k, 5: 98.7k, 6: 101.1k; GPU_MAX_HW_QUEUES=8 with 4 / 6 engines: 80.3k / 89.7k.
mkdir -p gpurun_out/r3
run() { name=$1; shift; v=$(env ZLY_BENCH_NO_H2H=1 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"); echo "$name $v"; }
B="timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --blocks 10 --no-extras --no-cpu-baseline"
run e3 $B --engines 3
run e2 $B --engines 2
run e4 $B --engines 4
run e3 $B --engines 3
run e5 $B --engines 5
run e6 $B --engines 6
run e4q8 GPU_MAX_HW_QUEUES=8 $B --engines 4
run e6q8 GPU_MAX_HW_QUEUES=8 $B --engines 6
run e3 $B --engines 3
