#!/bin/bash
# usage: ab_env.sh OUTDIR ENVNAME v1 v2 ...   (alternates values, 3 rounds; the value "-" = variable unset, for switches that only test presence)
out=$1; name=$2; shift 2
mkdir -p $out
for r in 1 2 3; do for v in "$@"; do
  t=$(basename "$v")                      # a value may be a path (ZLY_LIB=.../libzly_x.so): file names take its last component
  if [ "$v" = "-" ]; then setv="-u $name"; else setv="$name=$v"; fi
  env $setv timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/ab_${name}_${t}_$r.json 2>$out/ab.err || exit 1
  python3 -c "import json;d=json.load(open('$out/ab_${name}_${t}_$r.json'));print('$name=$v round $r',d['value'],d['ms_per_step'])"
done; done
