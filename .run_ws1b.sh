mkdir -p gpurun_out/r3
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ragged or taps or layer or yolov8s" > gpurun_out/r3/t_ws1b.log 2>&1
tail -5 gpurun_out/r3/t_ws1b.log
export AB_ROWS="model\.(6|8|9|12|18|21)\.cv"
bash zero-latency-yolo_amd/tools/ab.sh ws1b A=1 && bash zero-latency-yolo_amd/tools/ab.sh ws1b_nodual ZLY_WS1_NO_DUAL=1 && bash zero-latency-yolo_amd/tools/ab.sh ws1b_2 ZLY_WS1=2 && bash zero-latency-yolo_amd/tools/ab.sh ws1b_0 ZLY_WS1=0
for v in 1 0; do ZLY_WS1=$v ZLY_BENCH_NO_H2H=1 timeout -k 10 300 python3 bench.py --size 640 --batch 32 --scale s --steps 20 --warmup 5 --blocks 10 --no-cpu-baseline --per-launch-out gpurun_out/r3/pl_s640_ws1_$v.json > gpurun_out/r3/bench_s640_ws1_$v.json 2>gpurun_out/r3/bench_s640_ws1_$v.err; python3 -c "
import json;d=json.load(open('gpurun_out/r3/bench_s640_ws1_$v.json'));print('s640 ws1=$v',d['value'],d['ms_per_step'],d['roofline']['kernel_ms_per_step'])"; done
