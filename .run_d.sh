mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3/t_full_d.log 2>&1
tail -3 gpurun_out/r3/t_full_d.log
for i in 1 2; do
for lib in libzly.so libzly_base.so; do
ZLY_LIB=$PWD/zero-latency-yolo_amd/_build/$lib ZLY_BENCH_NO_H2H=1 timeout -k 10 200 python3 bench.py --batch 1 --engines 1 --steps 200 --warmup 20 --blocks 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib b1', d['value'], d['ms_per_step'])"
done; done
ZLY_BENCH_NO_H2H=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --blocks 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new', d['value'], d['ms_per_step'], d['latency_path_b1'])"
ZLY_LIB=$PWD/zero-latency-yolo_amd/_build/libzly_base.so ZLY_BENCH_NO_H2H=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --blocks 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('base', d['value'], d['ms_per_step'], d['latency_path_b1'])"
