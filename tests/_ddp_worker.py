"""Worker for tests/test_ddp_cpu.py: one rank of a world_size-2 gloo job.  Runs the data-parallel step of
medical_image_classification_amd.ddp_train (same wrap_ddp / setup_distributed the GPU path uses) on a tiny VSSM whose SS2D
blocks run the CPU oracle core, and checks DDP semantics."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import torch.nn as nn

from medical_image_classification_amd import medmamba as mm
from medical_image_classification_amd.ddp_train import setup_distributed, shard_indices, wrap_ddp
from oracle import ss2d_oracle


def build():
    torch.manual_seed(0)
    net = mm.VSSM(depths=[1, 1], dims=[16, 32], num_classes=3, drop_path_rate=0.0)
    ss2d_oracle.install(net)
    return net.train()


def main():
    torch.set_num_threads(2)
    distributed, rank, world, local_rank = setup_distributed("gloo")
    assert distributed and world == 2
    # global batch of 4 samples, DistributedSampler-style shards
    g = torch.Generator().manual_seed(123)
    X = torch.randn(4, 3, 32, 32, generator=g)
    Y = torch.randint(0, 3, (4,), generator=g)
    idx = shard_indices(4, rank, world, epoch=0)
    other = shard_indices(4, 1 - rank, world, epoch=0)
    assert sorted(idx + other) == [0, 1, 2, 3], "shards must partition the data set"
    net = build()
    if rank == 1:                       # DDP must broadcast rank 0's parameters at construction
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    ddp = wrap_ddp(net, True, local_rank, on_cuda=False)
    ref = build()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.equal(p, q), f"param {n} not broadcast from rank 0"
    lossf = nn.CrossEntropyLoss()
    loss = lossf(ddp(X[idx]), Y[idx])
    loss.backward()
    if os.environ.get("MEDSCAN_DDP", "flat") != "torch":
        from medical_image_classification_amd.ddp_train import FlatGradDataParallel
        assert isinstance(ddp, FlatGradDataParallel)
        assert len(ddp.slices) == 4 and ddp._next == -1, "every slice must have been launched from the backward hooks"
        ddp.reduce_gradients()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(ddp.params, ddp.views))
        with torch.no_grad():                                   # buffers: rank-local until sync_buffers()
            for b in net.buffers():
                if b.dtype.is_floating_point:
                    b.add_(float(rank))
        ddp.sync_buffers()
        chk = torch.cat([b.detach().flatten().float() for b in net.buffers()])
        got = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(got, chk)
        assert torch.equal(got[0], got[1]), "sync_buffers must leave rank 0's buffers on every rank"
    # (1) gradients are identical on all ranks after the all-reduce
    flat = torch.cat([p.grad.flatten() for p in net.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1]), "ranks disagree on the reduced gradient"
    # (2) they equal the mean over ranks of the local gradients (BatchNorm statistics are per-rank, as in the
    #     reference: no SyncBN, ddp_train.py:132-134), computed independently here
    mean = torch.zeros_like(flat)
    for r in range(world):
        m = build()
        ridx = shard_indices(4, r, world, epoch=0)
        lossf(m(X[ridx]), Y[ridx]).backward()
        mean += torch.cat([p.grad.flatten() for p in m.parameters()]) / world
    err = (flat - mean).abs().max().item()
    assert err <= 1e-5 * max(1.0, mean.abs().max().item()), f"DDP gradient != mean of local gradients ({err})"
    # (3) one optimizer step keeps the replicas in sync
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    opt.step()
    flat_p = torch.cat([p.detach().flatten() for p in net.parameters()])
    gp = [torch.zeros_like(flat_p) for _ in range(world)]
    dist.all_gather(gp, flat_p)
    assert torch.equal(gp[0], gp[1])
    # (4) a parameter that only ONE rank uses: every rank must step it with the same averaged gradient (torch DDP semantics);
    #     a parameter nobody uses keeps grad None.  The slices before the unused one are held back on the rank that does not
    #     use it and go out from reduce_gradients() in the same order as on the other rank.
    if os.environ.get("MEDSCAN_DDP", "flat") != "torch":
        class Two(nn.Module):
            def __init__(self):
                super().__init__()
                self.a, self.b, self.c, self.never = nn.Linear(4, 4), nn.Linear(4, 4), nn.Linear(4, 4), nn.Linear(4, 4)

            def forward(self, x, use_b):
                x = self.a(x)
                if use_b:
                    x = self.b(x)
                return self.c(x)
        torch.manual_seed(1)
        m = Two()
        w = FlatGradDataParallel(m, n_slices=4)
        o = torch.optim.SGD(m.parameters(), lr=0.1)
        x = torch.randn(3, 4, generator=torch.Generator().manual_seed(7 + rank))
        w.train()
        w(x, use_b=(rank == 0)).sum().backward()
        # rank 1 does not use b: its slice (and the ones in front of it) wait for reduce_gradients(); rank 0 has sent all but
        # the slice of the never-used parameter
        assert w._next >= 0
        w.reduce_gradients()
        assert m.b.weight.grad is not None, "a parameter used on another rank must receive the averaged gradient"
        assert m.never.weight.grad is None, "a parameter unused everywhere keeps grad None"
        o.step()
        fp = torch.cat([p.detach().flatten() for p in m.parameters()])
        gq = [torch.zeros_like(fp) for _ in range(world)]
        dist.all_gather(gq, fp)
        assert torch.equal(gq[0], gq[1]), "replicas diverged after a step with a rank-dependent unused parameter"
    dist.barrier()
    if rank == 0:
        print("DDP_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
