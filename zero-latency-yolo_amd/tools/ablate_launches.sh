#!/bin/bash
# What does each launch cost the STEP?  bench.py's headline configuration (three engines, batch 64) once as is, then once per launch with that
# launch left out (ZLY_ABLATE_SKIP, results garbage): step time saved = the launch's marginal cost with the other chains overlapping it.
# Output: gpurun_out/ablate.txt  (op | isolated us from profiles/r03_per_launch.json | ms/step without it | saved us)
out=gpurun_out/ablate; mkdir -p $out
# ZLY_ABLATE_SKIP exists in the diagnostic library only (make diaglib); the shipped libzly.so ignores it
export ZLY_LIB=$(pwd)/zero-latency-yolo_amd/_build/libzly_diag.so
[ -f "$ZLY_LIB" ] || { echo "build it first: make diaglib"; exit 1; }
export ZLY_BENCH_NO_H2H=1
run() { python3 bench.py --batch 64 --steps 20 --warmup 5 --blocks 12 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
base=$(run)
base2=$(run)
echo "baseline ms/step: $base $base2" > gpurun_out/ablate.txt
python3 -c "
import json
d=json.load(open('${PER_LAUNCH:-profiles/r03_per_launch.json}'))
for r in d['per_launch']: print(r['op'].split('+')[0].split(' ')[0] if False else r['op'], '|', r['us'])
" > $out/ops.txt
echo "stem | front-kernel" >> $out/ops.txt      # the front kernel (launched around the graph): 'model.0' skips nothing under that name
while IFS='|' read -r op us; do
  op=$(echo "$op" | sed 's/ *$//'); us=$(echo $us)
  t=$(ZLY_ABLATE_SKIP="$op" run)
  echo "$op | isolated $us us | without it $t ms/step | saved $(python3 -c "print(round(($base2+$base)/2*1000-$t*1000,1))") us" >> gpurun_out/ablate.txt
done < $out/ops.txt
cat gpurun_out/ablate.txt
