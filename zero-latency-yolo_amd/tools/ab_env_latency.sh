#!/bin/bash
name=$1; shift
for r in 1 2 3; do for v in "$@"; do
  if [ "$v" = "-" ]; then setv="-u $name"; else setv="$name=$v"; fi
  env $setv ZLY_BENCH_NO_H2H=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json;d=json.loads(sys.stdin.readline());l=d['latency_path_b1'];print('$name=$v round $r headline',d['value'],' b1 dev',l['ms_per_step_device_resident'],'p50',l['p50_detect_ms_host_to_host'],'p99',l['p99_detect_ms_host_to_host'])"
done; done
