#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/prof_$1/ (copy what is to be judged into profiles/):
#   kernel trace + stats of ONE engine's chain (--engines 1: per-position table, one-step timeline), kernel stats of the default
#   three-engine headline, PMC traffic (two passes, one engine)
set -e
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export ZLY_BENCH_NO_H2H=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt3 -- python3 bench.py --batch 64 --steps 40 --warmup 5 --blocks 2 --no-extras --no-cpu-baseline > $out/kt3_bench.json 2> $out/kt3.err
cp "$(find $out/kt3 -name '*kernel_stats.csv' | head -1)" $out/kernel_stats_3engines.csv
rm -rf $out/kt3
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --batch 64 --steps 40 --warmup 5 --blocks 2 --no-extras --no-cpu-baseline --engines 1 > $out/kt_bench.json 2> $out/kt.err
kt=$(find $out/kt -name "*kernel_trace.csv" | head -1)
st=$(find $out/kt -name "*kernel_stats.csv" | head -1)
cp "$st" $out/kernel_stats.csv
n=$(python3 - "$kt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_model1_kernel" in r["Kernel_Name"] or "stem_fused_kernel" in r["Kernel_Name"]]
print(idx[-2] - idx[-3])
PY
)
python3 zero-latency-yolo_amd/tools/trace_summary.py "$kt" $n $out/per_position.txt
python3 zero-latency-yolo_amd/tools/trace_timeline.py "$kt" 3 > $out/timeline.txt
# the latency path (batch 1, one engine): per-dispatch begin / end / gap of one step, and the kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt1 -- python3 bench.py --batch 1 --steps 200 --warmup 20 --blocks 2 --no-extras --no-cpu-baseline --engines 1 > $out/kt1_bench.json 2> $out/kt1.err
kt1=$(find $out/kt1 -name "*kernel_trace.csv" | head -1)
cp "$(find $out/kt1 -name '*kernel_stats.csv' | head -1)" $out/kernel_stats_b1.csv
python3 zero-latency-yolo_amd/tools/trace_timeline.py "$kt1" 5 > $out/timeline_b1.txt
rm -rf $out/kt1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_r -- python3 bench.py --batch 64 --steps 6 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline --engines 1 > /dev/null 2> $out/pmc_r.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_w -- python3 bench.py --batch 64 --steps 6 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline --engines 1 > /dev/null 2> $out/pmc_w.err
python3 zero-latency-yolo_amd/tools/pmc_traffic.py $out/pmc_r $out/pmc_w $out/traffic_b64.json
rm -rf $out/pmc_r $out/pmc_w $out/kt
echo "profile set in $out"
