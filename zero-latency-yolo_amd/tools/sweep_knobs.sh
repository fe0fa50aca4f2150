#!/bin/bash
# One box, one sitting: the headline configuration under a list of tuning-switch settings with baseline runs in between (noise: +-0.8 %).
# Round 3 result: nothing beyond noise -- stream grids of 768 / 1536 workgroups -2 %, ZLY_LDS_MIN_TILES=768 -6 %, ZLY_C2F_LDS_KB=120 -4 %.
run() { name=$1; shift; v=$(env ZLY_BENCH_NO_H2H=1 "$@" timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --blocks 10 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"); echo "$name $v"; }
run base A=1
run stream_wgs768 ZLY_STREAM_WGS=768
run stream_wgs1536 ZLY_STREAM_WGS=1536
run lds_wgs2 ZLY_LDS_WGS_PER_CU=2
run lds_wgs4 ZLY_LDS_WGS_PER_CU=4
run base A=1
run stream_min2048 ZLY_STREAM_MIN_GROUPS=2048
run stream_min16384 ZLY_STREAM_MIN_GROUPS=16384
run lds_min256 ZLY_LDS_MIN_TILES=256
run lds_min768 ZLY_LDS_MIN_TILES=768
run base A=1
run ws_min128 ZLY_WS_MIN_TILES=128
run ws_min1024 ZLY_WS_MIN_TILES=1024
run no_tail_split ZLY_NO_TAIL_SPLIT=1
run c2f_lds120 ZLY_C2F_LDS_KB=120
run base A=1
