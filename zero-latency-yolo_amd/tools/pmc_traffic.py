"""HBM-side traffic per batch step from two rocprofv3 counter passes (MI355X_MICROARCH.md, HBM / rocprofv3 section):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR_R -- python3 bench.py --batch 64 --steps 6 --warmup 1 --no-extras --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d DIR_W -- python3 bench.py ... (same command)
    python3 tools/pmc_traffic.py DIR_R DIR_W profiles/rNN_traffic_b64.json
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB per count as rocprofv3 reports them; gfx950 correction: FETCH_SIZE x 2.
Steps are counted by the nms_kernel dispatches (one per step); sums are over the MFMA conv family and over all kernels."""
import csv
import glob
import json
import os
import sys

CONV = ("conv_igemm_kernel", "conv3x3_lds_kernel", "conv3x3_ws_kernel", "stem_fused_kernel", "stem_model1_kernel", "conv1x1_stream_kernel", "bottleneck_pair_kernel", "c2f_kernel")


def collect(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    rows = []
    for f in files:
        for row in csv.DictReader(open(f, newline="")):
            if row.get("Counter_Name") == counter:
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], int(row["Grid_Size"]), float(row["Counter_Value"])))
    rows.sort()
    # the engine's warm-up runs at batch 1: keep only what follows the last nms_kernel dispatch that is not batch 64 (grid 64 x 512)
    cutoff = max([d for d, name, grid, _ in rows if "nms_kernel" in name and grid != 64 * 512], default=-1)
    conv = total = 0.0
    steps = 0
    for d, name, grid, v in rows:
        if d <= cutoff:
            continue
        total += v
        if any(k in name for k in CONV):
            conv += v
        if "nms_kernel" in name:
            steps += 1
    return conv, total, steps


def main():
    dr, dw, out = sys.argv[1:4]
    fr, fr_all, sr = collect(dr, "FETCH_SIZE")
    wr, wr_all, sw = collect(dw, "WRITE_SIZE")
    if sr < 1 or sw < 1:
        raise SystemExit("no nms_kernel dispatches found: cannot count steps")
    read_b, write_b = 2.0 * fr * 1024 / sr, wr * 1024 / sw
    j = {
        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes), python3 bench.py --batch 64 --steps 6 --warmup 1 --no-extras; "
                  "MFMA conv family (conv_igemm + conv3x3_lds + conv3x3_ws + conv1x1_stream + bottleneck_pair + c2f + stem_model1) summed over one batch-64 step, mean over the profiled steps",
        "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as read; both counters include Infinity-Cache hits",
        "steps_profiled": [sr, sw],
        "fetch_size_kb_per_step": fr / sr, "write_size_kb_per_step": wr / sw,
        "read_bytes_per_step": int(read_b), "write_bytes_per_step": int(write_b), "traffic_bytes_per_step": int(read_b + write_b),
        "all_kernels_traffic_bytes_per_step": int(2.0 * fr_all * 1024 / sr + wr_all * 1024 / sw),
        "batch": 64,
    }
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j))


if __name__ == "__main__":
    main()
