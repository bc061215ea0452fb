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


def _bench(args, **env):
    e = dict(os.environ, OMP_NUM_THREADS="2", PYTHONPATH=ROOT, **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=300)


def test_bench_self_launches_its_ranks_without_a_launcher():
    """`python bench.py --gpus N` as the driver may start it, with no torch.distributed.run around it: bench.py starts N fresh rank
    processes with the env:// rendezvous variables of ddp_train.py:64-81 and relays rank 0's ONE JSON line.  --dry-launch = the
    rendezvous, a barrier, one all-reduce and the JSON line without the model (gloo here; the GPU rehearsal runs the model)."""
    import json
    r = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch"], MEDSCAN_DIST_BACKEND="gloo")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                   # stdout carries exactly one line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] == 2.0 and j["backend"] == "gloo" and j["steps"] == 3 and j["warmup"] == 1


def test_bench_self_launch_fails_when_a_rank_fails():
    r = _bench(["--gpus", "2", "--dry-launch"], MEDSCAN_DIST_BACKEND="no_such_backend")
    assert r.returncode != 0
    assert not r.stdout.decode().strip()                            # no JSON line from a failed job
