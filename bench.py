#!/usr/bin/env python3
"""bench.py -- the detect path's headline benchmark (BASELINE.json: frames/sec + p50 detect latency,
YOLOv8n 416x416, batch 1 / 64, 1-8 MI355X).

    python bench.py --gpus 1 --steps K --warmup W            (default: N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (preprocess -> YOLOv8n forward -> decode -> NMS -> result slab)
over one batch of synthetic frames that are already resident in HBM.  By default the steps alternate over three engine
instances per GPU (--engines 3, ZLY_FLAG_SINGLE_CHAIN): each step is one chain of launches on its engine's stream and the chains
of consecutive steps overlap on the device; --engines 1 runs one engine with Detect side streams and deferred NMS.  BASELINE.json's metric is
"frames/sec + p50 detect latency, YOLOv8n 416x416 batch 1/64": the headline `value` is frames/s on the
batch-64 throughput configuration (configs[2]); the batch-1 latency configuration (configs[1]: frames/s
with resident frames, and the host-to-host p50 detect latency through zly_detect) is measured in the
same run and reported in the same JSON line as `latency_path_b1`; `throughput_host_to_host` is the same path fed
from host memory by >= 8 threads through zly_submit / zly_wait and through the plugin (native driver, child process).  With N > 1 each rank (one process per GPU) detects its own frames -- frames are sharded
one-per-GPU, no data-path collective -- and the per-frame result slabs are all-gathered over
RCCL/xGMI, overlapped with the next step (weak scaling: per-GPU work is fixed).

Everything measured goes through the C ABI of libzly.so; oracle/ is used only for the
`cpu_baseline` leg (the CPU restatement timed on the host cores, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in ("zero-latency-yolo_amd", "zero-latency-yolo_amd/tools"):
    sys.path.insert(0, os.path.join(ROOT, _p))

import shard          # noqa: E402
import zly            # noqa: E402
import zly_model as zm  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0         # HBM3E spec peak
ATTAINABLE_HBM_GBS = 6290.0   # the bandwidth the per-layer attainable time is priced against


def per_launch_roofline(ops, kernels, launches, ms, nb):
    """SURVEY 8d per launch: attainable_us = max(flops / peak_mfma, bytes / 6.29 TB/s) next to the achieved time.
    The engine fuses ops into launches depending on the batch size; zly_launch_info_at says which launch covers which ops and what its
    algorithmic work is.  Two byte counts per launch: `MB_unfused` = the sum of its ops' own algorithmic bytes (every op reads its input
    and writes its output once) and `MB` = fused-algorithmic = what has to cross the launch boundary (inputs not produced inside the launch +
    outputs read outside it + weights).  A launch is priced against `MB`; weights count once per launch, not per frame.
    frac = attainable / achieved <= 1 (1.0 = on its roofline)."""
    out = []
    for i, (o, k, li, m) in enumerate(zip(ops, kernels, launches, ms)):
        if li["covered_by"] != i or m <= 0:
            continue
        gf = li["flops_per_frame"] * nb / 1e9
        mb = ((li["bytes_fused_per_frame"] - li["weight_bytes"]) * nb + li["weight_bytes"]) / 1e6
        mbu = ((li["bytes_unfused_per_frame"] - li["weight_bytes"]) * nb + li["weight_bytes"]) / 1e6
        us = float(m) * 1e3
        t_mfma = gf * 1e9 / (PEAK_BF16_TFLOPS * 1e12) * 1e6
        t_hbm = mb * 1e6 / (ATTAINABLE_HBM_GBS * 1e9) * 1e6
        att = max(t_mfma, t_hbm)
        covered = [ops[j]["name"] for j, lj in enumerate(launches) if lj["covered_by"] == i and j != i]
        out.append({"op": o["name"], "covers": covered, "kernel": k, "kind": o["kind"], "us": round(us, 2), "gflop": round(gf, 3), "MB": round(mb, 2),
                    "MB_unfused": round(mbu, 2), "attainable_us": round(att, 2),
                    "bound": ("mfma" if t_mfma >= t_hbm else "hbm") if att > 0 else "latency",
                    "frac": round(att / us, 4), "TFLOPs": round(gf / us * 1e3, 1), "mfma_frac": round(gf / us * 1e3 / PEAK_BF16_TFLOPS, 4),
                    "GBps": round(mb / us * 1e3, 0), "hbm_frac": round(mb / us * 1e3 / PEAK_HBM_GBS, 4), "is_conv": o["kind"] == 1})
    return out


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


FORCE_GATHER = os.environ.get("ZLY_BENCH_FORCE_GATHER") == "1"


def run_steps(engs, frame_sets, batch, steps, slabs, stream_ptr, world, gather_out, size=416, k0=0, no_gather=False):
    """enqueue `steps` steps (global step numbers k0 .. k0+steps-1), step k on engine k % len(engs).
    One engine: it enqueues on the caller's stream and runs NMS of step k on its own stream beside the first kernels of step k+1
    (ZLY_FLAG_ASYNC_NMS).  Several engines (ZLY_FLAG_SINGLE_CHAIN): each step is one chain of launches on its engine's own stream and
    the chains of consecutive steps overlap -- what one chain leaves idle at its ~40 kernel boundaries and in its latency-bound small-map
    layers, the others fill.  With world > 1 the slabs of step k are all-gathered once step k+1 has been enqueued: zly_join orders the
    caller's stream behind step k's NMS, then the collective is queued -- overlapped with the following steps.
    Slab ring: len(slabs) should be a multiple of len(engs), so that a slab buffer is always written by the same engine (whose calls are
    stream-ordered among themselves); a buffer is rewritten only after the gather that read it has COMPLETED (an event behind the
    gather, waited for on the host -- it finished `ring` steps ago)."""
    import torch.distributed as dist
    n_eng = len(engs)
    ring = len(slabs)
    works, done = {}, {}
    gather = (world > 1 or FORCE_GATHER) and not no_gather      # no_gather: the same steps without the collective (the gather's share of a step)
    on_gpu = slabs[0].is_cuda

    def gather_step(j, lag):
        engs[j % n_eng].join(stream_ptr, lag)
        if j - 2 in works:
            works.pop(j - 2).wait()                    # gather buffer j%2 is free again once its previous gather finished
        works[j] = dist.all_gather_into_tensor(gather_out[j % 2], slabs[j % ring], async_op=True)
        if on_gpu:
            works[j].wait()                            # orders the caller's stream (joins + gathers only) behind the collective: no host wait
            done[j] = torch.cuda.Event()
            done[j].record()

    for k in range(k0, k0 + steps):
        d = frame_sets[k % len(frame_sets)]
        if k - ring in works:
            works.pop(k - ring).wait()                 # the slab buffer of step k - ring has been gathered
        if k - ring in done:
            done.pop(k - ring).synchronize()           # ... really: the engine's own stream is about to overwrite it
        engs[k % n_eng].detect_device(d.data_ptr(), batch, size, size, d_slabs_ptr=slabs[k % ring].data_ptr(), tag0=k * batch,
                                      stream=stream_ptr if n_eng == 1 else 0)
        if gather and k > k0:
            gather_step(k - 1, 1 if n_eng == 1 else 0)     # one engine: NMS(k-1), not NMS(k) -- step k+1 must not queue behind NMS(k)
    if gather and steps > 0:
        gather_step(k0 + steps - 1, 0)
    for w in works.values():
        w.wait()
    for e in engs:
        e.join(stream_ptr)                             # every engine's last NMS is ordered into the timed stream


def timed_blocks(engs, frame_sets, batch, steps, warmup, blocks, slabs, stream_ptr, world, gather_out, size=416, no_gather=False):
    """`warmup` untimed steps, then `blocks` consecutive blocks of EXACTLY `steps` steps, each bracketed by barrier +
    torch.cuda.synchronize() on both sides and timed with the maximum over ranks.  Returns the list of block times in seconds.
    Why blocks: the chip ramps its clock for the first ~50 steps after an idle phase (measured: 0.94 -> 0.72 ms/step over the first
    60 batch-64 steps, again after 2 s of idling; tools/first_steps.py) -- a single 20-step window measured right after 5 warm-up steps
    lies inside that ramp.  The median block is the steady state the metric is about; all block times are reported."""
    import torch.distributed as dist
    run_steps(engs, frame_sets, batch, warmup, slabs, stream_ptr, world, gather_out, size, k0=0, no_gather=no_gather)
    k0 = warmup
    out = []
    for _ in range(blocks):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(engs, frame_sets, batch, steps, slabs, stream_ptr, world, gather_out, size, k0=k0, no_gather=no_gather)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        out.append(dt)
        k0 += steps
    return out


def cpu_baseline(frames_np, seconds=12.0):
    """The CPU oracle (oracle/: C pre/post-processing + PyTorch-CPU fp32 forward on the same synthetic
    weights) timed on this host's cores.  kind = "port": ONNX Runtime, which the reference calls for the
    forward pass, is not available offline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import yolov8_ref
    from oracle_lib import Oracle
    orc = Oracle()
    ref = yolov8_ref.load(zly.DEFAULT_WEIGHTS, "fp32")

    def one(f):
        rc, x = orc.preprocess(f, 416, 416)
        head = ref.forward(torch.from_numpy(x[None])).numpy()[0]
        return orc.postprocess(head, f.shape[1], f.shape[0])

    out = {}
    for label, nthreads in (("all", host_cores()), ("2", 2)):
        torch.set_num_threads(max(1, nthreads))
        log(f"cpu_baseline: {nthreads} threads")
        for i in range(3):
            one(frames_np[i % len(frames_np)])
        lat = []
        t_end = time.perf_counter() + seconds / 2
        i = 0
        while time.perf_counter() < t_end or len(lat) < 10:
            t0 = time.perf_counter()
            one(frames_np[i % len(frames_np)])
            lat.append(time.perf_counter() - t0)
            i += 1
        out[label] = (len(lat) / sum(lat), float(np.median(lat)) * 1e3, len(lat), torch.get_num_threads())
    fps, p50, n, cores = out["all"]
    return {"value": round(fps, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "p50_ms": round(p50, 2),
            "sample": f"{n} single-frame 416x416 detects (oracle C preprocess + torch-CPU fp32 YOLOv8n + oracle decode/NMS), ~{seconds / 2:.0f} s on all cores + ~{seconds / 2:.0f} s on 2 threads",
            "value_2_threads": round(out["2"][0], 2), "p50_ms_2_threads": round(out["2"][1], 2),
            "reference_claim_fps": 60,
            "note": "ORT-CPU unavailable offline; torch-CPU stand-in for the forward pass. 60 FPS is the reference's unmeasured sleep-throttle target (README.md:16, onnx_engine.cpp:462-466)"}


def host_to_host(threads, seconds=4.0, max_batch=64, engines=1):
    """SURVEY 8d's throughput metric proper: request bytes in HOST memory -> detections in HOST memory, `threads` submitting host
    threads (config 3: >= 8).  Measured natively by _build/zly_h2h_bench (zero-latency-yolo_amd/tools/bench_h2h.cpp: no interpreter
    between the threads and the C ABI), once through zly_submit / zly_wait, once through HipInferenceEngine::submitInference ->
    InferenceCallback (what the reference's NetworkServer drives), and once as ONE client sending lone frames through the plugin
    (the latency a client of the server sees).  Frames are pageable host memory; the figures include the one host copy into the
    pinned ring, the PCIe upload (overlapped with the previous batch's compute) and the slab download."""
    import subprocess
    exe = os.path.join(ROOT, "zero-latency-yolo_amd", "_build", "zly_h2h_bench")
    out = {"threads": threads, "engines": engines, "pcie_ceiling_frames_per_sec": round(63e9 / 519168, 0)}
    for mode in ("cabi", "plugin", "lone"):
        try:
            args = [exe, zly.DEFAULT_WEIGHTS, mode, str(threads if mode != "lone" else 1), str(seconds if mode != "lone" else 2.0), str(max_batch),
                    str(engines if mode != "lone" else 1)]
            r = subprocess.run(args, capture_output=True, text=True, timeout=120)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out[mode] = json.loads(line[-1]) if (r.returncode == 0 and line) else {"error": (r.stderr or r.stdout)[-400:], "rc": r.returncode}
        except Exception as exc:       # noqa: BLE001  (a failing leg must not take the headline down)
            out[mode] = {"error": repr(exc)}
    return out


def self_launch(n):
    """bench.py --gpus N (N > 1) without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child
    and hand its one JSON line (rank 0's) through.  Exits non-zero when the node has fewer than N GPUs or the child fails."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"--gpus {n} but this node shows {have} GPU(s): refusing to print a line for a configuration that did not run")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise SystemExit(f"self-launched {n}-rank run failed (rc {r.returncode})")
    res = json.loads(lines[-1])
    if res.get("n_gpus") != n or res.get("rccl_ranks") != n:
        raise SystemExit(f"self-launched run reports n_gpus={res.get('n_gpus')} rccl_ranks={res.get('rccl_ranks')} under --gpus {n}")
    res["launcher"] = "self (bench.py started torch.distributed.run as a child process)"
    return json.dumps(res)


def main():
    # The driver reads ONE JSON line from stdout.  Libraries underneath print there too (RCCL writes its
    # version banner to stdout when the communicator is created), so everything but the final line is sent
    # to stderr: fd 1 is pointed at fd 2 for the whole run and the result goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = run()
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
    if line is not None:
        os.write(1, (line + "\n").encode())


def run():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="frames per step per GPU for the headline value (BASELINE configs[2] = 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the batch-64 / latency / roofline legs")
    ap.add_argument("--eager", action="store_true", help="no hipGraph replay")
    ap.add_argument("--keep-head", action="store_true", help="also materialise the fp32 [4+nc][N] head tensor (parity/debug output; the shipped plugin does not)")
    ap.add_argument("--size", type=int, default=416, help="square model/frame size; 640 = BASELINE configs[3] (then only the headline leg runs)")
    ap.add_argument("--scale", default="n", choices=["n", "s"], help="model scale; s + --fp8 + --size 640 = BASELINE configs[4] per GPU (then only the headline leg runs)")
    ap.add_argument("--fp8", action="store_true", help="weights stored as fp8 e4m3 (dequantised at load; the engine computes in bf16)")
    ap.add_argument("--engines", type=int, default=3, help="engine instances per GPU fed alternate steps (ZLY_FLAG_SINGLE_CHAIN when > 1); 1 = one engine with "
                                                          "side streams + deferred NMS")
    ap.add_argument("--sync-nms", action="store_true", help="run NMS in stream order at the end of every step instead of beside the next step's first kernels")
    ap.add_argument("--dump-ops", default="", help="write the per-launch hipEvent profile as a text table to this file")
    ap.add_argument("--blocks", type=int, default=0, help="timed blocks of --steps steps each (default: max(10, ceil(400 / steps))); the median block is reported")
    ap.add_argument("--per-launch-out", default=os.path.join(ROOT, "gpurun_out", "bench_per_launch.json"), help="where the per-launch roofline tables go (kept out of the JSON line)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" in os.environ and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the line would claim a GPU count it did not run on")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started bare with --gpus N > 1 (no launcher): this process has not touched the GPU yet (torch.cuda.device_count() does not
        # initialise it), so it becomes the launcher itself -- N fresh ranks under torch.distributed.run as CHILD processes, rank 0's JSON
        # line relayed, the child's exit code returned.  It never falls through to a one-GPU run that prints n_gpus = 1 under --gpus N.
        return self_launch(a.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    force_gather = os.environ.get("ZLY_BENCH_FORCE_GATHER") == "1"     # rehearse the RCCL path on one GPU
    if world > 1 or force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    B = a.batch
    big = 64
    default_cfg = a.size == 416 and a.scale == "n" and not a.fp8
    if not default_cfg:
        a.no_cpu_baseline = True      # the CPU leg and the native host-to-host driver are written for BASELINE's configs[1]/[2] (YOLOv8n 416 x 416);
                                      # the batch-1, latency and roofline legs run for every configuration
    wpath = None
    if a.scale != "n" or a.fp8:
        wpath = os.path.join(ROOT, "zero-latency-yolo_amd", "_build", f"yolov8{a.scale}_synth{'_fp8' if a.fp8 else ''}.zlyw")
        if rank == 0 and not os.path.exists(wpath):
            spec_ = zm.build_spec(a.scale)
            zm.write_zlyw(wpath, spec_, zm.synth_weights(spec_), fp8=a.fp8)
        if world > 1:
            dist.barrier()
    n_eng = max(1, a.engines)
    if n_eng > 1:
        eflags = (0 if a.keep_head else zly.FLAG_NO_HEAD_TENSOR) | zly.FLAG_SINGLE_CHAIN      # NMS in chain order: the other engines fill its tail
    else:
        eflags = (0 if a.keep_head else zly.FLAG_NO_HEAD_TENSOR) | (0 if a.sync_nms else zly.FLAG_ASYNC_NMS)
    engs = [zly.Engine(wpath, dtype=zly.DTYPE_BF16, model_w=a.size, model_h=a.size, max_batch=max(B, big), max_dets=64, device=local_rank, warmup_runs=3,
                       use_graph=not a.eager, flags=eflags) for _ in range(n_eng)]
    eng = engs[0]
    # a real (non-default) torch stream: the engine enqueues on it, and torch.distributed orders the RCCL
    # all-gather of the slabs behind it (with the legacy default stream the engine would fall back to its
    # own stream and the collective would not be ordered after NMS)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sp = stream.cuda_stream
    n_sets = 4
    frames_np = zm.synth_frames(n_sets * big, a.size, a.size, seed=20250328 + rank, rects=False)
    d_all = torch.from_numpy(frames_np).cuda()
    sets_b = [d_all[i * B:(i + 1) * B] for i in range(n_sets * big // B)][:64]
    sets_1 = [d_all[i:i + 1] for i in range(64)]
    sb = eng.slab_bytes

    def slab_bufs(n):
        return [torch.zeros(n * sb, dtype=torch.uint8, device="cuda") for _ in range(2 * n_eng if n_eng > 1 else 4)]   # a multiple of n_eng: see run_steps

    def gather_bufs(n):
        return [torch.zeros(world * n * sb, dtype=torch.uint8, device="cuda") for _ in range(2)] if (world > 1 or force_gather) else None

    n_blocks = a.blocks if a.blocks > 0 else max(10, -(-400 // max(1, a.steps)))
    log(f"engine ready (rank {rank}/{world}), headline leg: batch {B}, {a.warmup} warm-up steps, {n_blocks} blocks x {a.steps} steps")
    # ---- headline: BASELINE configs[2] (batch B per GPU per step) -----------------------------------
    gb_head = gather_bufs(B)
    slabs_head = slab_bufs(B)
    bt = timed_blocks(engs, sets_b, B, a.steps, a.warmup, n_blocks, slabs_head, sp, world, gb_head, a.size)
    dt = float(np.median(bt))
    value = world * B * a.steps / dt
    ms_per_step = dt / a.steps * 1e3
    result = {
        "metric": "frames_per_sec", "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 5),
        # what was timed: `blocks` consecutive blocks of `steps` steps, every block bracketed by barrier + synchronize; value = the MEDIAN block
        "blocks": n_blocks, "steps_timed": n_blocks * a.steps, "timed_region_s": round(float(sum(bt)), 5), "median_block_s": round(dt, 6),
        "mean_ms_per_step_all_blocks": round(float(sum(bt)) / (n_blocks * a.steps) * 1e3, 5),
        # ranks of the RCCL communicator the slab gather ran on (torch.distributed backend nccl = RCCL); 1 = no communicator, no collective
        "rccl_ranks": (dist.get_world_size() if dist.is_initialized() else 1),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"YOLOv8-{'nano' if a.scale == 'n' else 'small'} {a.size}x{a.size} batch={B} throughput path (BASELINE configs[2]), bf16{', fp8 e4m3 weight file' if a.fp8 else ''}",
                   "detail": f"per MI355X, frames resident in HBM, preprocess+forward+decode+NMS per step" + (f", steps alternate over {n_eng} engine instances whose chains overlap" if n_eng > 1 else "") +
                             (", slabs all-gathered over RCCL" if world > 1 else ""),
                   "engines_per_gpu": n_eng, "frames_per_step_per_gpu": B, "global_frames_per_step": world * B, "conf": 0.5, "iou": 0.45,
                   "weights": "seeded synthetic (no real weights offline)", "graph": not a.eager,
                   "parallelism": f"frame-sharded x{world}" if world > 1 else "single GPU",
                   "timing": f"median of {n_blocks} consecutive blocks of {a.steps} steps, each bracketed by barrier+sync (max over ranks)"},
        "blocks_ms_per_step": [round(t / a.steps * 1e3, 4) for t in bt],
        "first_block_ms_per_step": round(bt[0] / a.steps * 1e3, 4), "best_block_ms_per_step": round(min(bt) / a.steps * 1e3, 4),
    }
    lat_b1 = None
    if not a.no_extras:
        # ---- BASELINE configs[1]: batch 1, the latency path, frames resident in HBM ---------------------
        k1 = max(200, a.steps * 2)
        log(f"headline {value:.0f} frames/s (blocks ms/step: {result['blocks_ms_per_step']}); batch-1 leg")
        bt1 = timed_blocks(engs[:1], sets_1, 1, k1, max(20, a.warmup), 5, slab_bufs(1), sp, world, gather_bufs(1), a.size)      # the latency path: ONE in-order chain
        dt1 = float(np.median(bt1))
        lat_b1 = {"value": round(world * k1 / dt1, 1), "unit": "frames/s", "steps": k1, "ms_per_step_device_resident": round(dt1 / k1 * 1e3, 5)}
        if rank == 0:
            # ---- p50 detect latency, request bytes in host memory -> detections in host memory -----------
            log("latency leg")
            lat = []
            f_host = [np.ascontiguousarray(frames_np[i]) for i in range(16)]
            for i in range(50):
                eng.detect(f_host[i % 16])
            for i in range(400):
                t0 = time.perf_counter()
                eng.detect(f_host[i % 16])
                lat.append(time.perf_counter() - t0)
            lat = np.array(lat) * 1e3
            lat_b1.update({"p50_detect_ms_host_to_host": round(float(np.percentile(lat, 50)), 4),
                           "p90_detect_ms_host_to_host": round(float(np.percentile(lat, 90)), 4),
                           "p99_detect_ms_host_to_host": round(float(np.percentile(lat, 99)), 4), "samples": len(lat),
                           "path": f"zly_detect: {a.size * a.size * 3 // 1000} KB H2D over PCIe + path + slab D2H, synchronous"})
            # ---- roofline of the dominant kernel family (the MFMA conv launches of one forward) ----------
            log("roofline leg (per-op hipEvent profile)")
            ops = eng.ops()
            roof, tables = {}, {}
            # Two per-launch measures, both with hipEvents on the engine's stream (zly_profile_ops):
            #  * one launch per event pair -- includes the few microseconds an event pair adds, as rocprofv3's
            #    per-dispatch durations do (the committed profiles/*_kernel_stats.csv agree with THIS figure);
            #  * 8 launches back to back per event pair, time / 8 -- the in-stream cost of a launch.
            # `achieved` uses the first (conservative, rocprof-consistent); the second is reported beside it.
            for nb, frames in ((B, sets_b[0]), (1, sets_1[0])):
                os.environ["ZLY_PROFILE_INNER"] = "8"
                ms8 = eng.profile_ops(frames.data_ptr(), nb, a.size, a.size, reps=10)
                os.environ["ZLY_PROFILE_INNER"] = "1"
                ms = eng.profile_ops(frames.data_ptr(), nb, a.size, a.size, reps=20)
                rows = per_launch_roofline(ops, eng.op_kernels(nb), eng.launches(nb), ms, nb)
                conv_l = [r for r in rows if r["is_conv"]]
                dom = max(conv_l, key=lambda r: r["us"])
                conv_us = sum(r["us"] for r in conv_l)
                conv8_ms = float(sum(m for o, m in zip(ops, ms8) if o["kind"] == 1))
                gflop = sum(r["gflop"] for r in conv_l)
                mb_f = sum(r["MB"] for r in conv_l)
                mb_u = sum(r["MB_unfused"] for r in conv_l)
                tfl = gflop / conv_us * 1e3
                gbs = mb_f / conv_us * 1e3
                # the family's bound = the bound that most of its kernel time sits under in the per-launch model
                t_hbm = sum(r["us"] for r in conv_l if r["bound"] == "hbm")
                bound = "hbm" if t_hbm * 2 >= conv_us else "mfma"
                att = sum(r["attainable_us"] for r in rows)
                att_conv = sum(r["attainable_us"] for r in conv_l)
                roof[nb] = {"bound": bound,
                            "achieved": round(gbs if bound == "hbm" else tfl, 1), "peak": PEAK_HBM_GBS if bound == "hbm" else PEAK_BF16_TFLOPS,
                            "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                            "frac": round((gbs / PEAK_HBM_GBS) if bound == "hbm" else (tfl / PEAK_BF16_TFLOPS), 5), "traffic": None,
                            "kernel": "all MFMA conv launches of one forward (stem_model1 / c2f / conv3x3_lds / conv1x1_stream / conv_igemm / bottleneck_pair)",
                            "bound_rule": f"{t_hbm / conv_us:.0%} of the conv kernel time is in launches whose per-launch model bound is hbm",
                            "launches_per_step": len(conv_l), "kernel_ms_per_step": round(conv_us / 1e3, 4), "avg_launch_us": round(conv_us / len(conv_l), 2),
                            "algorithmic_gflop_per_step": round(gflop, 2), "algorithmic_GB_per_step_fused": round(mb_f / 1e3, 4),
                            "algorithmic_GB_per_step_unfused": round(mb_u / 1e3, 4),
                            "mfma_achieved_tflops": round(tfl, 1), "mfma_frac": round(tfl / PEAK_BF16_TFLOPS, 5),
                            "hbm_achieved_GBps": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 5),
                            "back_to_back_kernel_ms_per_step": round(conv8_ms, 4), "back_to_back_mfma_frac": round(gflop / conv8_ms / PEAK_BF16_TFLOPS, 5),
                            "all_launches_ms_per_step": round(sum(r["us"] for r in rows) / 1e3, 4),
                            "attainable_ms_per_step": round(att / 1e3, 4), "attainable_conv_ms_per_step": round(att_conv / 1e3, 4),
                            "achieved_over_attainable": round(sum(r["us"] for r in rows) / max(1e-9, att), 2),
                            "dominant_op": dom["op"], "dominant_kernel": dom["kernel"], "dominant_us": dom["us"], "dominant_bound": dom["bound"],
                            "dominant_GBps": dom["GBps"], "dominant_hbm_frac": dom["hbm_frac"], "dominant_TFLOPs": dom["TFLOPs"], "dominant_mfma_frac": dom["mfma_frac"],
                            "dominant_frac_of_attainable": dom["frac"],
                            "other_launches_us": {r["op"]: r["us"] for r in rows if not r["is_conv"]},
                            "method": "zly_profile_ops: 20 eager passes on the engine's stream, one hipEvent pair per launch; bytes = fused-algorithmic (zly_launch_info_at)"}
                tables[nb] = [{k: v for k, v in r.items() if k != "is_conv"} for r in rows]
                if a.dump_ops:
                    with open(a.dump_ops, "a") as f:
                        f.write(f"# batch {nb}: per-launch mean us over 20 eager reps (hipEvents around every launch); MB = fused-algorithmic\n")
                        f.write(f"{'op':40s} {'kernel':46s} {'us':>8s} {'GFLOP':>8s} {'MB':>8s} {'MBunf':>8s} {'att_us':>7s} {'frac':>6s} {'TFLOP/s':>8s} {'GB/s':>7s} bound\n")
                        for r in rows:
                            f.write(f"{r['op'][:40]:40s} {r['kernel'][:46]:46s} {r['us']:8.2f} {r['gflop']:8.3f} {r['MB']:8.2f} {r['MB_unfused']:8.2f} {r['attainable_us']:7.2f} "
                                    f"{r['frac']:6.3f} {r['TFLOPs']:8.1f} {r['GBps']:7.0f} {r['bound']}\n")
                        f.write(f"{'TOTAL':40s} {'':46s} {sum(r['us'] for r in rows):8.2f}\n\n")
            # the per-launch tables go to a file (the JSON line stays small enough for the driver to keep all of it)
            try:
                os.makedirs(os.path.dirname(a.per_launch_out), exist_ok=True)
                with open(a.per_launch_out, "w") as f:
                    json.dump({"batch": B, "per_launch": tables[B], "per_launch_b1": tables[1] if B != 1 else None,
                               "note": "per launch: us = hipEvent time; MB = fused-algorithmic bytes, MB_unfused = sum of the covered ops' own bytes; attainable_us = max(gflop / 2.5 PF, MB / 6.29 TB/s); frac = attainable / achieved"}, f, indent=1)
                roof[B]["per_launch_file"] = os.path.relpath(a.per_launch_out, ROOT)
            except OSError as exc:
                roof[B]["per_launch_file"] = f"not written: {exc}"
            bad = [r["op"] for r in tables[B] if r["frac"] > 1.0]
            roof[B]["launches_with_frac_above_1"] = bad
            # HBM-side traffic of the same kernels from the PMC counters (collected off-line with rocprofv3, two
            # --pmc passes, gfx950 correction applied; see the file for provenance): per batch-64 step
            for tp in ("r04_traffic_b64.json", "r03_traffic_b64.json", "r02_traffic_b64.json"):
                tpath = os.path.join(ROOT, "profiles", tp)
                if B == 64 and default_cfg and os.path.exists(tpath):
                    tj = json.load(open(tpath))
                    roof[B]["traffic"] = tj["traffic_bytes_per_step"]
                    roof[B]["traffic_over_fused_algorithmic"] = round(tj["traffic_bytes_per_step"] / (roof[B]["algorithmic_GB_per_step_fused"] * 1e9), 3)
                    roof[B]["traffic_note"] = "bytes per step from profiles/" + tp + " (rocprofv3 --pmc, FETCH_SIZE doubled per the gfx950 note, WRITE_SIZE as read)"
                    break
            roof[B]["whole_step_tflops"] = round(roof[B]["algorithmic_gflop_per_step"] / ms_per_step, 1)
            roof[B]["whole_step_mfma_frac"] = round(roof[B]["algorithmic_gflop_per_step"] / ms_per_step / PEAK_BF16_TFLOPS, 4)
            result["roofline"] = roof[B]
            if B != 1:
                r1 = roof[1]
                result["roofline_b1"] = {k: r1[k] for k in ("bound", "launches_per_step", "kernel_ms_per_step", "avg_launch_us", "mfma_frac", "hbm_frac",
                                                            "back_to_back_kernel_ms_per_step", "all_launches_ms_per_step", "attainable_ms_per_step")}
            hdrs = [h for h, _ in eng.read_slabs(min(B, 64))]
            result["candidates_per_frame"] = {"median": float(np.median([int(h["n_candidates"]) for h in hdrs])),
                                              "max": int(max(int(h["n_candidates"]) for h in hdrs)), "kept_max": int(max(int(h["n_kept"]) for h in hdrs))}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(frames_np[:8])
        result["speedup_vs_cpu_baseline"] = round(value / result["cpu_baseline"]["value"], 1)
    st = eng.stats()
    result["engine_stats"] = {"inference_count": st["inference_count"], "inference_errors": st["inference_errors"]}
    if world > 1 or force_gather:
        # the gathered slabs of the last headline step, back in global frame order, must be well-formed, and this rank's slice of them must
        # be, byte for byte, what ONE engine writes for the same frames in a plain synchronous call
        last = a.warmup + n_blocks * a.steps - 1
        g = shard.global_order(gb_head[last % 2], world * B, world, sb)
        hd = g[:, :16].contiguous().view(torch.int32).cpu().numpy()
        assert (hd[:, 0] >= 0).all() and (hd[:, 1] >= hd[:, 0]).all(), "malformed gathered slabs"
        ref_slab = torch.zeros(B * sb, dtype=torch.uint8, device="cuda")
        eng.detect_device(sets_b[last % len(sets_b)].data_ptr(), B, a.size, a.size, d_slabs_ptr=ref_slab.data_ptr(), tag0=last * B, stream=sp if n_eng == 1 else 0)
        eng.join(sp)
        torch.cuda.synchronize()
        mine = gb_head[last % 2].view(world, B * sb)[rank].view(B, sb).cpu().numpy()
        want = ref_slab.view(B, sb).cpu().numpy()
        cap_d = (sb - 16) // 40
        for f in range(B):
            # a slab = 16-byte header + cap detections of which the first n_kept are written: the bytes behind them are whatever an earlier
            # step left in the ring buffer, so the comparison covers the header and the written detections
            nk = min(int(want[f, :4].view(np.int32)[0]), cap_d)
            assert np.array_equal(mine[f, :16 + 40 * nk], want[f, :16 + 40 * nk]), f"gathered slab of frame {f} of the last step differs from a single-engine run on the same frames"
        result["gather"] = {"ranks": world, "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "bytes_per_rank_per_step": B * sb,
                            "frames_checked": int(hd.shape[0]), "equals_single_engine_run": True}
        # the gather's share of a step: the same steps once more WITHOUT the collective (3 blocks, same bracket, max over ranks)
        bt_ng = timed_blocks(engs, sets_b, B, a.steps, 0, 3, slabs_head, sp, world, gb_head, a.size, no_gather=True)
        ms_ng = float(np.median(bt_ng)) / a.steps * 1e3
        result["gather"]["ms_per_step_without_gather"] = round(ms_ng, 5)
        result["gather"]["share_of_step"] = round(max(0.0, 1.0 - ms_ng / ms_per_step), 4)
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_extras and default_cfg and os.environ.get("ZLY_BENCH_NO_H2H") != "1":
        for e_ in engs:                # the native driver creates its own engines: free this process's first
            e_.close()
        engs = []
        log("host-to-host legs (native driver: C ABI, plugin, lone client)")
        h2h = host_to_host(max(8, min(12, host_cores() - 3)), engines=2)      # 2 engines + the shared upload stream = one hardware queue per stream
        keep = ("frames_per_sec", "pcie_h2d_GBps", "p50_ms", "p99_ms", "avg_batch", "errors", "error")
        result["throughput_host_to_host"] = {"threads": h2h["threads"], "engines": h2h["engines"], "pcie_ceiling_frames_per_sec": h2h["pcie_ceiling_frames_per_sec"],
                                             "cabi": {k: v for k, v in h2h.get("cabi", {}).items() if k in keep},
                                             "plugin": {k: v for k, v in h2h.get("plugin", {}).items() if k in keep}}
        if "frames_per_sec" in h2h.get("cabi", {}):
            result["throughput_host_to_host"]["frac_of_device_resident"] = round(h2h["cabi"]["frames_per_sec"] / value, 3)
        if lat_b1 is not None and "p50_ms" in h2h.get("lone", {}):
            lo = h2h["lone"]
            lat_b1["plugin_lone_client"] = {"p50_ms": lo["p50_ms"], "p90_ms": lo["p90_ms"], "p99_ms": lo["p99_ms"], "frames_per_sec": lo["frames_per_sec"],
                                            "path": "one client, HipInferenceEngine::submitInference -> InferenceCallback, lone frames (batch-1 graph replayed by the pipelined path)"}
        elif lat_b1 is not None:
            lat_b1["plugin_lone_client"] = h2h.get("lone")
    for e_ in engs:
        e_.close()
    if world > 1 or force_gather:
        dist.destroy_process_group()
    if lat_b1 is not None:
        result["latency_path_b1"] = lat_b1          # BASELINE's "p50 detect latency": LAST, so that it is in the tail of the line whatever is cut
    return json.dumps(result) if rank == 0 else None


if __name__ == "__main__":
    main()
