#!/bin/bash
# per-kernel SQ counters of the batch-64 step (one rocprofv3 pass per counter group, --pmc with --kernel-trace only); prints per-kernel means
out=gpurun_out/pmc_k; mkdir -p $out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export ZLY_BENCH_NO_H2H=1
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 bench.py --batch 64 --steps 4 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline --engines 1 > /dev/null 2> $out/p$i.err || echo "group $i failed: $grp"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)            # kernel -> dispatch durations in ns (the counter rows carry the dispatch's timestamps)
for f in glob.glob("gpurun_out/pmc_k/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        if int(r["Grid_Size"]) < 64 * 256: continue       # skip the batch-1 warm-up launches
        name = r["Kernel_Name"].replace("void zly::", "").replace("zly::", "")[:46]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r.get("Start_Timestamp") and r.get("End_Timestamp") and r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            dur[name].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
ctrs = sorted({c for k in acc.values() for c in k})
with open("gpurun_out/pmc_k/summary.txt", "w") as o:
    o.write("kernel".ljust(48) + " ".join(c.replace("SQ_", "")[:14].rjust(15) for c in ctrs) + "\n")
    for k, v in sorted(acc.items()):
        o.write(k.ljust(48) + " ".join(("%15.4g" % (sum(v[c]) / len(v[c])) if c in v else " " * 15) for c in ctrs) + "\n")
    # MFMA utilisation per kernel (the north-star's "MFMA utilisation against gfx950 peak"): SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a
    # SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs; peak = every SIMD busy for the whole dispatch at the 2.4 GHz maximum clock
    o.write("\nmfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (dispatch duration x 2.4 GHz x 1024 SIMDs)   [profiled dispatches, mean]\n")
    for k, v in sorted(acc.items()):
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and dur.get(k):
            busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(v["SQ_VALU_MFMA_BUSY_CYCLES"])
            d = sum(dur[k]) / len(dur[k])
            o.write(k.ljust(48) + "%10.1f us  busy %12.4g  mfma_util %6.3f\n" % (d / 1e3, busy, busy / (d * 2.4 * 1024)))
print(open("gpurun_out/pmc_k/summary.txt").read())
PY
rm -rf $out/p?
