"""Synthetic-data version of the reference's fusion training loop (CrossMamba/train.py:73-166): VFEFM over
two modalities, Adam, learning rate x0.75 per epoch after the first, output clamped to [0,1], `Fusionloss`.  The reference
loop needs cv2 and the CT/MRI dataset; here the two inputs are random images in [0,1] of the same shape.

    python -m medical_image_classification_amd.train_fusion --batch-size 4 --res 224 --steps 10 --bf16
"""
import argparse
import time

import torch

from .crossmamba import VFEFM
from .fusion_loss import FusionLoss
from .train import make_adam


def build_fusion_model(small=False, **kw):
    """The model CrossMamba/train.py:80-91 builds (`small`: a 4-stage toy of the same topology for tests)."""
    if small:
        cfg = dict(depths=[1, 1, 1, 1], dims=[64, 128, 256, 512], depths_decoder=[1, 1, 1, 1], dims_decoder=[512, 256, 128, 64],
                   d_state=16)
    else:
        cfg = dict(depths=[2, 2, 4, 2], dims=[128, 256, 512, 1024], depths_decoder=[2, 9, 2, 2], dims_decoder=[1024, 512, 256, 128])
    cfg.update(kw)
    return VFEFM(in_chans=3, patch_size=4, cat_method="stack", attn_drop_rate=0.0, drop_path_rate=0.1, use_checkpoint=False, **cfg)


def synthetic_pair(batch, res, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(batch, 3, res, res, generator=g).to(device), torch.rand(batch, 3, res, res, generator=g).to(device))


def fusion_step(model, opt, criterion, vis, ir, autocast_dtype=None):
    """One iteration of CrossMamba/train.py:120-135."""
    if autocast_dtype is not None:
        with torch.autocast("cuda", dtype=autocast_dtype):
            fused = model(vis, ir)
    else:
        fused = model(vis, ir)
    fused = fused.float().clamp(0, 1)                      # the two torch.where of train.py:126-129
    opt.zero_grad(set_to_none=True)
    total, loss_in, ssim_value, loss_grad = criterion(vis, ir, fused)
    total.backward()
    opt.step()
    return total.detach(), loss_in.detach(), ssim_value.detach(), loss_grad.detach()


def epoch_lr(base_lr, epoch):
    return base_lr if epoch == 0 else base_lr * (0.75 ** (epoch - 1))     # train.py:115


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="iterations per epoch")
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--small", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = build_fusion_model(small=a.small).to(dev).train()
    opt = make_adam(model.parameters(), lr=a.lr)
    crit = FusionLoss().to(dev)
    vis, ir = synthetic_pair(a.batch_size, a.res, dev)
    for epo in range(a.epochs):
        for g in opt.param_groups:
            g["lr"] = epoch_lr(a.lr, epo)
        for it in range(a.steps):
            t0 = time.perf_counter()
            total, loss_in, ssim_value, loss_grad = fusion_step(model, opt, crit, vis, ir, torch.bfloat16 if a.bf16 else None)
            torch.cuda.synchronize()
            print(f"epoch {epo} step {it + 1}/{a.steps} loss_total {total.item():.4f} loss_in {loss_in.item():.4f} "
                  f"loss_grad {loss_grad.item():.4f} ssim_loss {ssim_value.item():.4f} time {time.perf_counter() - t0:.3f}s", flush=True)


if __name__ == "__main__":
    main()
