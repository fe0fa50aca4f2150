mkdir -p gpurun_out/r3
export AB_ROWS="model\.(12|18)\.cv"
bash zero-latency-yolo_amd/tools/ab.sh c_base A=1 && bash zero-latency-yolo_amd/tools/ab.sh c_m2 ZLY_WS1=2 && bash zero-latency-yolo_amd/tools/ab.sh c_m2_76 ZLY_WS1=2 ZLY_WS1_LDS_KB=76 && bash zero-latency-yolo_amd/tools/ab.sh c_76 ZLY_WS1_LDS_KB=76 && bash zero-latency-yolo_amd/tools/ab.sh c_base2 A=1
for cfg in "1 64" "2 64" "2 76" "1 76"; do set -- $cfg; ZLY_WS1=$1 ZLY_WS1_LDS_KB=$2 ZLY_BENCH_NO_H2H=1 timeout -k 10 300 python3 bench.py --size 640 --batch 32 --scale s --steps 20 --warmup 5 --blocks 10 --no-cpu-baseline --per-launch-out gpurun_out/r3/pl_s640_c_$1_$2.json > gpurun_out/r3/bench_s640_c_$1_$2.json 2>gpurun_out/r3/bench_s640_c_$1_$2.err; python3 -c "
import json;d=json.load(open('gpurun_out/r3/bench_s640_c_$1_$2.json'));print('s640 ws1=$1 lds=$2',d['value'],d['ms_per_step'],d['roofline']['kernel_ms_per_step'])"; done
