"""N > 1 path on CPU: frame sharding + slab all-gather + global re-ordering over torch.distributed with
the gloo backend, world_size 2 (and 3, ragged).  Slabs are produced by the CPU oracle here (no GPU in
this container); on the MI355X node the same host logic runs over RCCL with slabs from the HIP path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import shard
import zly


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _slab_for(frame_id, cap):
    """deterministic fake detections for global frame `frame_id`"""
    rng = np.random.default_rng(1000 + frame_id)
    n = int(rng.integers(0, cap + 1))
    raw = np.zeros(16 + cap * 40, dtype=np.uint8)
    hdr = raw[:16].view(zly.SLAB_HDR_DTYPE)
    hdr["n_kept"] = n; hdr["n_candidates"] = n + 3; hdr["frame_tag"] = frame_id
    dets = raw[16:].view(zly.DET_DTYPE)
    for k in range(n):
        dets[k] = (*rng.uniform(0, 1, 4).astype(np.float32), np.float32(rng.uniform(0.5, 1)), int(rng.integers(0, 80)), 0, 0, 0)
    return raw


def _worker(rank, world, port, n_frames, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sb = 16 + cap * 40
    per = shard.frames_per_rank(n_frames, world)
    mine = shard.local_frame_ids(n_frames, world, rank)
    local = np.zeros((per, sb), dtype=np.uint8)                     # ranks with fewer frames pad with empty slabs
    for slot, fid in enumerate(mine):
        local[slot] = _slab_for(fid, cap)
    gathered, work = shard.gather_slabs(torch.from_numpy(local.reshape(-1)), world, async_op=True)
    work.wait()
    ordered = shard.global_order(gathered, n_frames, world, sb).numpy()
    ok = all(np.array_equal(ordered[i], _slab_for(i, cap)) for i in range(n_frames))
    tags = [int(h["frame_tag"]) for h, _ in zly.parse_slabs(ordered.reshape(-1), n_frames, cap)]
    q.put((rank, ok, tags, mine))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 8), (2, 5), (3, 7)])
def test_shard_gather_reorder(world, n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, 6, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = []
    for rank, ok, tags, mine in res:
        assert ok and tags == list(range(n_frames))                 # every rank sees every frame, in global order
        assert all(i % world == rank for i in mine)                 # one-frame-per-GPU round robin
        owned += mine
    assert sorted(owned) == list(range(n_frames))                   # each frame detected exactly once


def test_single_rank_is_identity():
    x = torch.arange(3 * 56, dtype=torch.uint8)
    g, w = shard.gather_slabs(x, 1)
    assert w is None and torch.equal(shard.global_order(g, 3, 1, 56).reshape(-1), x)
