#!/bin/bash
# The round's closing run on one box: GPU tests, the profile set (profile_round.sh, pmc_kernels.sh, batch-1 stats), then the driver's bench command and the
# other configurations with the fresh traffic figure in place.  Everything lands in gpurun_out/final/ (copy what is to be judged into profiles/).
tag=${1:-r03}
out=gpurun_out/final; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; tail -2 $out/gpu_tests.log
bash zero-latency-yolo_amd/tools/profile_round.sh $tag > $out/profile_round.log 2>&1 && cp gpurun_out/prof_$tag/* $out/ && cp gpurun_out/prof_$tag/traffic_b64.json profiles/${tag}_traffic_b64.json
bash zero-latency-yolo_amd/tools/pmc_kernels.sh > $out/pmc_kernels.txt 2> $out/pmc_kernels.err
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ZLY_BENCH_NO_H2H=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_b1 -- python3 bench.py --batch 1 --engines 1 --steps 200 --warmup 20 --blocks 3 --no-extras --no-cpu-baseline > $out/kt_b1_bench.json 2> $out/kt_b1.err
cp "$(find $out/kt_b1 -name '*kernel_stats.csv' | head -1)" $out/kernel_stats_b1.csv; rm -rf $out/kt_b1
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --per-launch-out $out/per_launch.json > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err
export ZLY_BENCH_NO_H2H=1
timeout -k 10 300 python3 bench.py --size 640 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --per-launch-out $out/per_launch_yolov8n_640_b32.json > $out/bench_yolov8n_640_b32.json 2> $out/b640n.err
timeout -k 10 300 python3 bench.py --size 640 --batch 32 --scale s --steps 20 --warmup 5 --no-cpu-baseline --per-launch-out $out/per_launch_yolov8s_640_b32.json > $out/bench_yolov8s_640_b32.json 2> $out/b640s.err
timeout -k 10 300 python3 bench.py --size 640 --batch 32 --scale s --fp8 --steps 20 --warmup 5 --no-cpu-baseline --per-launch-out $out/per_launch_yolov8s_640_b32_fp8.json > $out/bench_yolov8s_640_b32_fp8weights.json 2> $out/b640f.err
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
python3 - <<'PY'
import json
for n in ("bench_driver_cmd", "bench_yolov8n_640_b32", "bench_yolov8s_640_b32", "bench_yolov8s_640_b32_fp8weights"):
    try:
        d = json.load(open(f"gpurun_out/final/{n}.json")); r = d["roofline"]
        print(n, d["value"], d["ms_per_step"], "conv_ms", r["kernel_ms_per_step"], "frac", r["frac"], "mfma", r["mfma_frac"], "b1", d.get("latency_path_b1", {}).get("ms_per_step_device_resident"), d.get("latency_path_b1", {}).get("p50_detect_ms_host_to_host"))
    except Exception as e: print(n, "FAILED", e)
PY
