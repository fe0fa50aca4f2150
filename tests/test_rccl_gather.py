"""The slab exchange of the multi-GPU path on the real backend (RCCL via torch.distributed "nccl"), on the one GPU the
test box has: world 1, but the same call sequence bench.py queues at N > 1 (deferred NMS, zly_join(lag = 1), async
all_gather_into_tensor overlapped with the next step, shard.global_order).  Runs in a child process because the process
group must be created before any other GPU call of the process.  No scaling curve can be measured here (one GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_gather_of_real_engine_slabs():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_gather_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL gather ok" in r.stdout
