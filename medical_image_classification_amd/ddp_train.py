"""Data-parallel training -- counterpart of the reference's ddp_train.py (ddp_train.py:64-202): one process per
GPU, `init_process_group('nccl')` (= RCCL over xGMI on ROCm), DistributedSampler-style sharding of the batch
stream, gradient all-reduce overlapped with backward (default: `FlatGradDataParallel`, a few large slices of one flat
buffer launched from backward hooks; `MEDSCAN_DDP=torch`: torch's DistributedDataParallel reducer as the reference wraps
its model), a synthetic validation pass per epoch, rank-0 checkpoints `{epoch, model, optimizer, best_acc}` written when
the validation accuracy improves, and `--resume`.  Synthetic data replaces ImageFolder.

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
    """Data-parallel wrapper: the gradients live in ONE flat fp32 buffer that is all-reduced in a few large slices,
    each launched from backward as soon as its last gradient exists -- overlapped with the rest of backward like the
    reference's DDP reducer (ddp_train.py:134), without that reducer's per-parameter cost.

    torch's DDP pays per parameter: every backward it copies each of the 355 gradients into its bucket (a launch each;
    autograd hands over freshly allocated gradient tensors, so `gradient_as_bucket_view` cannot avoid it), runs its hooks and
    bucket bookkeeping, and broadcasts the BatchNorm buffers before every forward.  Measured on one MI355X with a ONE-rank
    process group (MEDSCAN_FORCE_DDP=1; nothing here has run on more than one GPU yet): 27.8 ms per step against 24.9 ms
    without a wrapper, 25.25 ms with this one.

    * The flat buffer is cut into `n_slices` contiguous slices at parameter boundaries (registration order = forward order,
      so backward completes them last slice first).  A post-accumulate hook per parameter counts its slice down; when a
      slice is complete its gradients are packed with one multi-tensor copy and `all_reduce(async_op=True)` is issued:
      RCCL runs it on its own stream behind the copy while backward continues on the compute stream.
    * Collectives must be issued in the same order on every rank: slices go out strictly last -> first.  A slice that is
      not complete when its turn comes (a parameter unused on THIS rank this step never fires its hook) holds back the ones
      before it; `reduce_gradients()` -- called between backward() and optimizer.step() (train.train_step does) -- issues
      whatever is left in the same order, waits for all of them and scales by 1 / world.
    * Unused parameters: a parameter contributes zeros where its gradient is None, and one flag per parameter travels at the
      end of the flat buffer: a parameter that ANY rank used gets the averaged gradient on EVERY rank (as under torch DDP);
      only a parameter unused everywhere keeps `grad = None`.  Without the flags the ranks that did not use it would skip
      its optimizer update and the replicas would drift apart silently.
    * Parameters and buffers are broadcast from rank 0 at construction, as DDP does; BatchNorm running statistics then stay
      rank-local (they never enter the training arithmetic; `sync_buffers()` puts rank 0's on every rank -- ddp_train.main
      calls it before each checkpoint / validation -- which is the state torch DDP's per-forward broadcast converges to)."""

    def __init__(self, module, n_slices=4):
        super().__init__()
        self.module = module
        self.world = dist.get_world_size()
        self._broadcast([p.data for p in module.parameters()] + [b.data for b in module.buffers()])
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params or any(p.dtype != torch.float32 or p.device != self.params[0].device for p in self.params):
            raise RuntimeError("FlatGradDataParallel needs fp32 parameters on one device (use MEDSCAN_DDP=torch otherwise)")
        sizes = [p.numel() for p in self.params]
        total, P = sum(sizes), len(self.params)
        # [gradients | one "used" flag per parameter]
        self.flat = torch.zeros(total + P, device=self.params[0].device, dtype=torch.float32)
        self.views = [v.view_as(p) for v, p in zip(self.flat[:total].split(sizes), self.params)]
        self.flags = self.flat[total:]
        # contiguous slices of about equal size, cut at parameter boundaries
        n_slices = max(1, min(int(os.environ.get("MEDSCAN_DDP_SLICES", n_slices)), P))
        self.slice_of, self.slices, acc, lo, first = [], [], 0, 0, 0
        for i, n in enumerate(sizes):
            self.slice_of.append(len(self.slices))
            acc += n
            if acc >= total * (len(self.slices) + 1) / n_slices or i == P - 1:
                self.slices.append((first, i + 1, lo, acc))          # parameters [first, i+1), flat elements [lo, acc)
                first, lo = i + 1, acc
        self._left = [0] * len(self.slices)
        self._next = len(self.slices) - 1                            # next slice to go out (last -> first)
        self._works = []
        self._armed = False
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

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
        # arm the slice counters for the backward of this forward (training only)
        if torch.is_grad_enabled() and self.module.training:
            if self._works:
                # a backward whose slices went out but were never reduced (gradient accumulation without reduce_gradients()):
                # the in-place collectives may still be writing the flat buffer the next backward packs into
                raise RuntimeError("FlatGradDataParallel: the previous backward's all-reduces are still outstanding -- call "
                                   "reduce_gradients() between backward() and the next forward (one all-reduce round per "
                                   "forward/backward; gradient accumulation over several forwards is not supported)")
            self._left = [hi - lo for lo, hi, _, _ in self.slices]
            self._next = len(self.slices) - 1
            self._works = []
            self._armed = True
        return self.module(*args, **kwargs)

    def sync_buffers(self):
        bufs = [b.data for b in self.module.buffers()]
        if bufs:
            self._broadcast(bufs)

    def _make_hook(self, i):
        def hook(_param):
            if not self._armed:
                return
            k = self.slice_of[i]
            self._left[k] -= 1
            while self._next >= 0 and self._left[self._next] == 0:      # in order, last slice first
                self._launch(self._next)
                self._next -= 1
        return hook

    def _launch(self, k):
        """Pack slice k (zeros where this rank has no gradient) and start its all-reduce."""
        first, last, lo, hi = self.slices[k]
        src, dst, missing = [], [], []
        for p, v in zip(self.params[first:last], self.views[first:last]):
            if p.grad is None:
                missing.append(v)
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad); dst.append(v)
        if missing:
            torch._foreach_zero_(missing)
        if src:
            torch._foreach_copy_(dst, src)
        self._works.append(dist.all_reduce(self.flat[lo:hi], async_op=True))

    def reduce_gradients(self):
        """Call between backward() and optimizer.step(): p.grad <- mean over ranks (train.train_step does)."""
        if not self._armed:                 # backward of a forward made without arming (eval / no_grad): plain path
            self._left = [0] * len(self.slices)
            self._next = len(self.slices) - 1
            self._works = []
        while self._next >= 0:              # slices held back by a parameter this rank did not use
            self._launch(self._next)
            self._next -= 1
        local = [p.grad is not None for p in self.params]
        all_local = all(local)
        if all_local:
            self.flags.fill_(1.0)           # no host data: nothing here makes the host wait for the device
        else:
            self.flags.copy_(torch.tensor([1.0 if u else 0.0 for u in local]))
        self._works.append(dist.all_reduce(self.flags, async_op=True))      # every rank contributes its flags, always
        for w in self._works:
            w.wait()                        # stream-level for RCCL: the compute stream waits, the host does not
        self._works, self._armed = [], False
        total = self.flat.numel() - len(self.params)
        self.flat[:total].mul_(1.0 / self.world)
        # A parameter this rank used is used "somewhere": when this rank used ALL of them (the normal case) the reduced flags
        # need not be read at all -- reading them is a device-to-host copy that would stall the host once per step, and with
        # ~770 kernels to enqueue per step the GPU then idles while the host catches up.
        used = local if all_local else [f > 0 for f in self.flags.tolist()]
        for p, v, f in zip(self.params, self.views, used):
            if f:
                p.grad = v                  # used on some rank: every rank steps it with the same averaged gradient
            # unused everywhere: grad stays None, the optimizer skips it on all ranks alike


def wrap_ddp(net, distributed, local_rank, on_cuda=True, broadcast_buffers=True):
    """The data-parallel wrapper of ddp_train.py:134.  Default: FlatGradDataParallel (sliced flat all-reduce overlapped with backward);
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
    ap.add_argument("--local_rank", type=int, default=None,
                    help="device index when LOCAL_RANK is not in the environment (torch.distributed.launch style); torchrun's "
                         "LOCAL_RANK wins")
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
    if args.local_rank is not None and "LOCAL_RANK" not in os.environ:
        os.environ["LOCAL_RANK"] = str(args.local_rank)
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
        # validation: EVERY rank evaluates the (unsharded) validation set, as in the reference (ddp_train.py:171-181: the val loader
        # has no DistributedSampler); rank 0 logs and saves on a strict improvement (`>`, ddp_train.py:186).  Synthetic batch, drawn
        # from one generator seed on all ranks so that they evaluate the same images.
        net.eval()
        with torch.no_grad():
            vgen = torch.Generator(device=device).manual_seed(4321 + epoch)
            images, labels = synthetic_batch(args.batch_size, args.num_classes, args.res, device, vgen)
            acc = (net(images).argmax(dim=1) == labels).sum().item() / args.batch_size
        if rank == 0:
            print(f"[epoch {epoch + 1}] train_loss: {running / args.steps_per_epoch:.3f}  val_accuracy: {acc:.3f}")
            if acc > best_acc:                  # checkpoint on improvement (ddp_train.py:186-194)
                best_acc = acc
                torch.save({"epoch": epoch, "model": net.state_dict(), "optimizer": optimizer.state_dict(),
                            "best_acc": best_acc}, args.save_path)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
