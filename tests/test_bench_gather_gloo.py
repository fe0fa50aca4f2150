"""bench.py's step loop at N > 1 on CPU: run_steps with several (fake) engines per rank, the slab ring, zly_join ordering calls and the
overlapped all-gather, over torch.distributed gloo at world size 2.  The fake engine writes the slabs a real one would leave in HBM
(frame tags = global step * batch + slot), so the gathered, re-ordered bytes of the last steps can be checked on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeEngine:
    """stands in for zly.Engine: detect_device fills the slab buffer whose address it is given"""

    def __init__(self, registry, cap, rank):
        self.registry, self.cap, self.rank = registry, cap, rank
        self.slab_bytes = 16 + cap * 40
        self.calls, self.joins = [], []

    def detect_device(self, d_frames_ptr, n, w, h, d_slabs_ptr=0, tag0=0, stream=0):
        buf = self.registry[d_slabs_ptr]
        raw = buf.numpy().reshape(n, self.slab_bytes)
        raw[:] = 0
        hdr = raw[:, :16].view(np.int32)
        for i in range(n):
            hdr[i, 0] = (tag0 + i + self.rank) % 5          # n_kept
            hdr[i, 1] = hdr[i, 0] + 1                        # n_candidates
            hdr[i, 3] = tag0 + i                             # frame_tag
        self.calls.append((tag0, stream))

    def join(self, stream=0, lag=0):
        self.joins.append(lag)


def _worker(rank, world, port, n_eng, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    import shard
    batch, steps, cap = 4, 11, 3
    registry = {}
    engs = [FakeEngine(registry, cap, rank) for _ in range(n_eng)]
    sb = engs[0].slab_bytes
    slabs = [torch.zeros(batch * sb, dtype=torch.uint8) for _ in range(2 * n_eng if n_eng > 1 else 4)]      # as bench.py: a multiple of n_eng
    for t in slabs:
        registry[t.data_ptr()] = t
    gather_out = [torch.zeros(world * batch * sb, dtype=torch.uint8) for _ in range(2)]
    frames = [torch.zeros(batch * 12, dtype=torch.uint8)]
    bench.run_steps(engs, frames, batch, steps, slabs, 1234, world, gather_out)
    ok = True
    for k in (steps - 1, steps - 2):                           # the two gather buffers hold the last two steps
        g = shard.global_order(gather_out[k % 2], world * batch, world, sb).numpy()
        hdr = g[:, :16].copy().view(np.int32)
        for i in range(world * batch):
            r, slot = i % world, i // world
            ok = ok and hdr[i, 3] == k * batch + slot and hdr[i, 0] == (k * batch + slot + r) % 5
    order = sorted((tag0, ei) for ei, e in enumerate(engs) for tag0, _ in e.calls)
    q.put((rank, bool(ok), [ei for _, ei in order], [e.joins for e in engs], [e.calls[0][1] for e in engs]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_eng", [1, 3])
def test_bench_step_loop_gathers_every_step_in_order(n_eng):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_eng, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, order, joins, streams in res:
        assert ok, rank
        assert order == [k % n_eng for k in range(11)]           # steps alternate over the engines
        if n_eng == 1:
            assert streams == [1234] and joins[0][:-2] == [1] * 10     # one engine: the caller's stream, gather of step k-1 behind NMS(k-1) only
        else:
            assert streams == [0] * n_eng and all(l == 0 for j in joins for l in j)   # several: own streams, join on that engine's last call


def test_bench_gpus_n_without_launcher_never_prints_a_one_gpu_line():
    """`bench.py --gpus 8` started bare (no WORLD_SIZE) used to run on one GPU and print n_gpus = 1.  Now it either becomes
    the launcher of N ranks itself or exits non-zero; on a box with fewer GPUs than asked for (here: none) it must fail WITHOUT a JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout
    assert "--gpus 8" in r.stderr
    # a launcher whose world size disagrees with --gpus is refused as well (in both directions)
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env2, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
    env3 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, env=env3, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()


def _worker_nogather(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    registry = {}
    engs = [FakeEngine(registry, 3, rank) for _ in range(3)]
    sb = engs[0].slab_bytes
    slabs = [torch.zeros(4 * sb, dtype=torch.uint8) for _ in range(6)]
    for t in slabs:
        registry[t.data_ptr()] = t
    gather_out = [torch.full((world * 4 * sb,), 7, dtype=torch.uint8) for _ in range(2)]
    bench.run_steps(engs, [torch.zeros(48, dtype=torch.uint8)], 4, 9, slabs, 1234, world, gather_out, no_gather=True)
    q.put((rank, all(bool((g == 7).all()) for g in gather_out), sum(len(e.calls) for e in engs)))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_step_loop_without_gather_leg():
    """the `gather.share_of_step` leg: the same steps with the collective left out touch no gather buffer and still run every step"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_nogather, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, untouched, calls in res:
        assert untouched and calls == 9, rank
