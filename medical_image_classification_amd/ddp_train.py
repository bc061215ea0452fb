"""Data-parallel training -- counterpart of the reference's ddp_train.py (ddp_train.py:64-202): one process per
GPU, `init_process_group('nccl')` (= RCCL over xGMI on ROCm), DistributedSampler-style sharding of the batch
stream, DistributedDataParallel with bucketed gradient all-reduce overlapped with backward, rank-0 checkpoints
`{epoch, model, optimizer, best_acc}` and `--resume`.  Synthetic data replaces ImageFolder.

Deliberate fixes relative to the reference: LOCAL_RANK is read from the environment
(the reference leaves `--local_rank` at 0, so under torchrun every rank picks cuda:0, ddp_train.py:56,74-75), and
resume loads into the unwrapped module (the reference loads a non-`module.` state_dict into the DDP wrapper,
ddp_train.py:142-148).
"""
import argparse
import os

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel as DDP

from .medmamba import BranchStreamTuner
from .train import build_model, make_adam, synthetic_batch, train_step


def setup_distributed(backend=None):
    """env:// rendezvous from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*; single process if they are unset
    (ddp_train.py:77-81).  Returns (distributed, rank, world_size, local_rank)."""
    # MEDSCAN_FORCE_DDP=1: build the process group and the DDP wrapper even for one rank (measures the wrapper's own cost)
    force = os.environ.get("MEDSCAN_FORCE_DDP") == "1"
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and (int(os.environ["WORLD_SIZE"]) > 1 or force):
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        local_rank = int(os.environ.get("LOCAL_RANK", rank))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("MEDSCAN_DIST_BACKEND") or backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local_rank %= max(1, torch.cuda.device_count())       # (rehearsals with more ranks than visible GPUs)
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        return True, rank, world, local_rank
    return False, 0, 1, int(os.environ.get("LOCAL_RANK", 0))


class FlatGradDataParallel(nn.Module):
    """Data-parallel wrapper with ONE gradient all-reduce per step over a flat fp32 buffer.

    torch's DDP reducer pays per parameter: every backward it copies each of the 355 gradients into its bucket (a launch
    each; autograd hands over freshly allocated gradient tensors, so `gradient_as_bucket_view` cannot avoid it), runs its
    hooks and bucket bookkeeping, and broadcasts the BatchNorm buffers before every forward.  Measured on one MI355X with a
    one-rank process group (MEDSCAN_FORCE_DDP=1): 27.1 ms per step against 24.9 ms without the wrapper -- an 8 % tax that
    every N > 1 run would pay before a byte crosses xGMI.  MedMamba-T has only 57.8 MB of gradients (about 0.5 ms of
    all-reduce on the xGMI mesh), so here: after backward the gradients are packed into one flat buffer with a multi-tensor
    copy (a handful of launches), all-reduced ONCE (RCCL picks its own chunking for a message this size), scaled by
    1/world, and handed to the optimizer as views of that buffer.  Same result as DDP (mean of the ranks' gradients).
    Parameters and buffers are broadcast from rank 0 at construction, as DDP does; BatchNorm running statistics then stay
    rank-local (they never enter the training arithmetic; `sync_buffers()` puts rank 0's on every rank -- ddp_train.main
    calls it before each checkpoint -- which is the state torch DDP's per-forward broadcast converges to)."""

    def __init__(self, module):
        super().__init__()
        self.module = module
        self.world = dist.get_world_size()
        self._broadcast([p.data for p in module.parameters()] + [b.data for b in module.buffers()])
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params or any(p.dtype != torch.float32 or p.device != self.params[0].device for p in self.params):
            raise RuntimeError("FlatGradDataParallel needs fp32 parameters on one device (use MEDSCAN_DDP=torch otherwise)")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), device=self.params[0].device, dtype=torch.float32)
        self.views = [v.view_as(p) for v, p in zip(self.flat.split([p.numel() for p in self.params]), self.params)]

    @staticmethod
    def _broadcast(tensors, src=0):
        from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors
        by_type = {}
        for t in tensors:
            by_type.setdefault((t.dtype, t.device), []).append(t)
        for group in by_type.values():
            flat = _flatten_dense_tensors(group)
            dist.broadcast(flat, src)
            for t, f in zip(group, _unflatten_dense_tensors(flat, group)):
                t.copy_(f)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def sync_buffers(self):
        bufs = [b.data for b in self.module.buffers()]
        if bufs:
            self._broadcast(bufs)

    def reduce_gradients(self):
        """Call between backward() and optimizer.step(): p.grad <- mean over ranks (train.train_step does)."""
        src, dst, missing = [], [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                missing.append(v)                      # (unused this step: contributes zeros, as under DDP)
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad); dst.append(v)
        if missing:
            torch._foreach_zero_(missing)
        if src:
            torch._foreach_copy_(dst, src)
        dist.all_reduce(self.flat)
        self.flat.mul_(1.0 / self.world)
        for p, v in zip(self.params, self.views):
            if p.grad is not None:
                p.grad = v


def wrap_ddp(net, distributed, local_rank, on_cuda=True, broadcast_buffers=True):
    """The data-parallel wrapper of ddp_train.py:134.  Default: FlatGradDataParallel (one flat all-reduce per step);
    MEDSCAN_DDP=torch: torch's DistributedDataParallel as the reference uses it (25 MiB buckets overlapped with backward,
    `gradient_as_bucket_view`, buffers broadcast every forward)."""
    if not distributed:
        return net
    if os.environ.get("MEDSCAN_DDP", "flat") != "torch":
        return FlatGradDataParallel(net)
    if on_cuda:
        return DDP(net, device_ids=[local_rank], output_device=local_rank, gradient_as_bucket_view=True,
                   broadcast_buffers=broadcast_buffers)
    return DDP(net, gradient_as_bucket_view=True, broadcast_buffers=broadcast_buffers)


def shard_indices(n_samples, rank, world, epoch, seed=0):
    """DistributedSampler semantics (ddp_train.py:111,153-154): shuffle with seed+epoch, pad to a multiple of
    world, rank r takes indices r::world."""
    g = torch.Generator().manual_seed(seed + epoch)
    idx = torch.randperm(n_samples, generator=g).tolist()
    total = -(-n_samples // world) * world
    idx += idx[: total - n_samples]
    return idx[rank:total:world]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--local_rank", type=int, default=0)
    ap.add_argument("--steps-per-epoch", type=int, default=10)
    ap.add_argument("--num-classes", type=int, default=8)
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--variant", default="T", choices=["T", "B", "SSD"],
                    help="T/B: MedMamba.py VSSM sizes; SSD: CNN_Mamba.py VSSM (what the reference's train.py imports)")
    ap.add_argument("--save-path", default="./MedmambaNet_ddp.pth")
    ap.add_argument("--resume", default="")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("ddp_train.py needs MI355X GPUs: the SS2D kernels have no CPU fallback")
    distributed, rank, world, local_rank = setup_distributed("nccl")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    net = build_model(num_classes=args.num_classes, variant=args.variant).to(device)
    start_epoch, best_acc = 0, 0.0
    optimizer = make_adam(net.parameters(), lr=0.0001)
    if args.resume:
        ckpt = torch.load(args.resume, map_location=device, weights_only=True)
        net.load_state_dict(ckpt["model"]); optimizer.load_state_dict(ckpt["optimizer"])
        start_epoch, best_acc = ckpt["epoch"] + 1, ckpt["best_acc"]
    ddp_net = wrap_ddp(net, distributed, local_rank)
    loss_function = nn.CrossEntropyLoss()
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    tuner = BranchStreamTuner(device)           # two-stream blocks: measured per process on the first steps (rank-local)
    for epoch in range(start_epoch, args.epochs):
        ddp_net.train()
        running = 0.0
        for _ in range(args.steps_per_epoch):
            images, labels = synthetic_batch(args.batch_size, args.num_classes, args.res, device, gen)
            tuner.begin()
            loss = train_step(ddp_net, optimizer, loss_function, images, labels, torch.bfloat16 if args.bf16 else None)
            tuner.end()
            running += loss.item()
        if hasattr(ddp_net, "sync_buffers"):
            ddp_net.sync_buffers()              # rank 0's BatchNorm statistics everywhere before they are saved / evaluated
        if rank == 0:
            print(f"[epoch {epoch + 1}] train_loss: {running / args.steps_per_epoch:.3f}")
            torch.save({"epoch": epoch, "model": net.state_dict(), "optimizer": optimizer.state_dict(),
                        "best_acc": best_acc}, args.save_path)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
