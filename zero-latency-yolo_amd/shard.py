"""Frame sharding across GPUs and the result-slab gather.

One process per GPU.  Frames are independent (the reference holds no cross-frame state in
runInference, onnx_engine.cpp:518-646), so global frame i goes to rank i % world -- the north-star's
one-frame-per-GPU granularity -- with NO data-path collective.  The only exchange step is after NMS:
every rank contributes its fixed-size result slabs and one all-gather (RCCL over xGMI when the backend
is "nccl"; gloo on CPU in the tests) gives every rank, in particular the dispatcher on rank 0, all
detections, which are then put back into global frame order.

Pure host logic over torch.distributed; no oracle, no kernels.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def local_frame_ids(n_frames: int, world: int, rank: int) -> List[int]:
    """global indices of the frames rank `rank` detects (round robin)."""
    return list(range(rank, n_frames, world))


def frames_per_rank(n_frames: int, world: int) -> int:
    """slabs every rank contributes to the gather (ranks with fewer frames pad with empty slabs)."""
    return (n_frames + world - 1) // world


def gather_slabs(local_slabs: torch.Tensor, world: int, out: torch.Tensor = None, async_op: bool = False):
    """local_slabs: u8 [per_rank * slab_bytes] on this rank's device.  Returns (gathered u8
    [world * per_rank * slab_bytes], work handle or None)."""
    if out is None:
        out = torch.empty(world * local_slabs.numel(), dtype=torch.uint8, device=local_slabs.device)
    if world == 1:
        out.copy_(local_slabs)
        return out, None
    work = dist.all_gather_into_tensor(out, local_slabs, async_op=async_op)
    return out, work


def global_order(gathered: torch.Tensor, n_frames: int, world: int, slab_bytes: int) -> torch.Tensor:
    """[world][per_rank][slab_bytes] (rank-major, as gathered) -> [n_frames][slab_bytes] in global frame
    order: frame i is slot i // world of rank i % world."""
    per = frames_per_rank(n_frames, world)
    g = gathered.view(world, per, slab_bytes)
    idx_rank = torch.arange(n_frames, device=gathered.device) % world
    idx_slot = torch.arange(n_frames, device=gathered.device) // world
    return g[idx_rank, idx_slot]
