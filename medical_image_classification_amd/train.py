"""Single-device training loop -- counterpart of the reference's train.py (train.py:14-109): VSSM + Adam(1e-4) +
CrossEntropyLoss, batch 32, best-accuracy state_dict checkpoint -- driven by SYNTHETIC data (the reference's
ImageFolder dataset is not available offline; shapes and the hot loop are the same)."""
import argparse
import os
import time

import torch
import torch.nn as nn

from .medmamba import VSSM as medmamba
from .medmamba import BranchStreamTuner


def synthetic_batch(batch_size, num_classes, res=224, device="cuda", generator=None):
    """`images, labels` with the shapes the reference's DataLoader yields (train.py:66-71)."""
    images = torch.randn(batch_size, 3, res, res, device=device, generator=generator)
    labels = torch.randint(0, num_classes, (batch_size,), device=device, generator=generator)
    return images, labels


def build_model(num_classes=8, variant="T", **kw):
    if variant == "SSD":                    # what the reference's train.py:11,58 builds as shipped: CNN_Mamba.VSSM
        from .cnn_mamba import VSSM as ssd_vssm
        return ssd_vssm(num_classes=num_classes, **kw)
    if variant == "B":                      # BASELINE.json config 3
        kw = dict(depths=[2, 2, 12, 2], dims=[128, 256, 512, 1024], **kw)
    return medmamba(num_classes=num_classes, **kw)


def make_adam(params, lr=0.0001):
    """`optim.Adam(net.parameters(), lr=0.0001)` of the reference (train.py:62).  On the GPU: `adam.MsAdam`, a torch.optim.Adam whose
    step updates all parameters in ONE launch (ms_adam_multi; same update rule, same state_dict; torch's fused implementation
    needs 8 under-filled launches, 0.34 vs 0.1 ms per step).  MEDSCAN_ADAM=torch: torch's fused Adam."""
    params = list(params)
    if params and all(p.is_cuda for p in params):
        if os.environ.get("MEDSCAN_ADAM", "ms") != "torch":
            from .adam import MsAdam
            return MsAdam(params, lr=lr)
        try:
            return torch.optim.Adam(params, lr=lr, fused=True)
        except (RuntimeError, TypeError, ValueError):
            pass
    return torch.optim.Adam(params, lr=lr)


def train_step(net, optimizer, loss_function, images, labels, autocast_dtype=None):
    """The hot loop body of train.py:73-77: zero_grad -> forward -> loss -> backward -> step."""
    optimizer.zero_grad(set_to_none=True)
    if autocast_dtype is not None:
        with torch.autocast(device_type="cuda", dtype=autocast_dtype):
            loss = loss_function(net(images), labels)
    else:
        loss = loss_function(net(images), labels)
    loss.backward()
    if hasattr(net, "reduce_gradients"):           # ddp_train.FlatGradDataParallel: one flat all-reduce per step
        net.reduce_gradients()
    optimizer.step()
    return loss


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--steps-per-epoch", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--num-classes", type=int, default=8)
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--variant", default="T", choices=["T", "B", "SSD"],
                    help="T/B: MedMamba.py VSSM sizes; SSD: CNN_Mamba.py VSSM (what the reference's train.py imports)")
    ap.add_argument("--save-path", default="./MedmambaNet.pth")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("train.py needs an MI355X: the SS2D kernels have no CPU fallback")
    device = torch.device("cuda:0")
    print(f"using {device} device.")
    net = build_model(num_classes=args.num_classes, variant=args.variant).to(device)
    loss_function = nn.CrossEntropyLoss()
    optimizer = make_adam(net.parameters(), lr=0.0001)
    gen = torch.Generator(device=device).manual_seed(0)
    best_acc = 0.0
    tuner = BranchStreamTuner(device)           # two-stream blocks: measured on the first steps, kept only if faster here
    for epoch in range(args.epochs):
        net.train()
        running_loss, t0 = 0.0, time.time()
        for step in range(args.steps_per_epoch):
            images, labels = synthetic_batch(args.batch_size, args.num_classes, args.res, device, gen)
            tuner.begin()
            loss = train_step(net, optimizer, loss_function, images, labels, torch.bfloat16 if args.bf16 else None)
            tuner.end()
            running_loss += loss.item()
        dt = time.time() - t0
        net.eval()
        acc = 0
        with torch.no_grad():
            images, labels = synthetic_batch(args.batch_size, args.num_classes, args.res, device, gen)
            acc = (net(images).argmax(dim=1) == labels).sum().item() / args.batch_size
        print(f"[epoch {epoch + 1}] train_loss: {running_loss / args.steps_per_epoch:.3f}  val_accuracy: {acc:.3f}  "
              f"{args.steps_per_epoch * args.batch_size / dt:.1f} images/s")
        if acc > best_acc:                      # strict improvement, as train.py:101
            best_acc = acc
            torch.save(net.state_dict(), args.save_path)
    print("Finished Training")


if __name__ == "__main__":
    main()
