"""N > 1 path on CPU: world_size 2, gloo backend, launched exactly like the driver launches bench.py
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...`)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


import pytest


@pytest.mark.parametrize("impl", ["flat", "torch"])
def test_ddp_world2_gloo(impl):
    """Both data-parallel wrappers (the flat single all-reduce, default; torch DDP as the reference uses it)."""
    env = dict(os.environ, OMP_NUM_THREADS="2", PYTHONPATH=ROOT, MEDSCAN_DDP=impl)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_ddp_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert "DDP_OK" in out, out[-3000:]


def test_shard_indices_semantics():
    from medical_image_classification_amd.ddp_train import shard_indices
    for n, w in ((10, 4), (7, 2), (8, 8), (3, 4)):
        shards = [shard_indices(n, r, w, epoch=3) for r in range(w)]
        assert len({len(s) for s in shards}) == 1                      # equal length (padded)
        assert set(sum(shards, [])) == set(range(n))                   # every sample is seen
    a = [shard_indices(64, r, 4, epoch=0) for r in range(4)]
    b = [shard_indices(64, r, 4, epoch=1) for r in range(4)]
    assert a != b                                                       # reshuffled per epoch (set_epoch)


def test_single_process_fallback(monkeypatch):
    from medical_image_classification_amd.ddp_train import setup_distributed, wrap_ddp
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    distributed, rank, world, local_rank = setup_distributed("gloo")
    assert (distributed, rank, world, local_rank) == (False, 0, 1, 0)
    import torch.nn as nn
    m = nn.Linear(2, 2)
    assert wrap_ddp(m, False, 0) is m
