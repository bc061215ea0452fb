"""Same scan problem (MedMamba-T stage shapes, bs 64) in the three addressing modes of the kernels:
BDL (reference (B,4D,L) layout), CL (channel-last, no map), SS2D (channel-last + pixel maps, shared u)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd import selective_scan_fn
from medical_image_classification_amd.ss2d_fused import _SS2DScan

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = 5
for D, Hh, R in [(96, 56, 3), (192, 28, 6), (384, 14, 12), (768, 7, 24)]:
    L, N, C = Hh * Hh, 16, R + 32
    gen = torch.Generator(device=dev).manual_seed(0)
    A = (-0.5 * torch.rand(4 * D, N, device=dev, generator=gen)).requires_grad_()
    Dp = torch.randn(4 * D, device=dev, generator=gen).requires_grad_()
    bias = (0.5 * torch.rand(4 * D, device=dev, generator=gen)).requires_grad_()
    def timeit(fn, g):
        out = fn(); out.backward(g); torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(iters):
            e[0].record(); out = fn(); e[1].record(); out.backward(g); e[2].record(); torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        return tf / iters, tb / iters
    # BDL
    u = torch.randn(bs, 4 * D, L, device=dev, generator=gen).requires_grad_()
    dl = (0.5 * torch.rand(bs, 4 * D, L, device=dev, generator=gen)).requires_grad_()
    xdbl = torch.randn(bs, 4, C, L, device=dev, generator=gen).requires_grad_()
    g = torch.randn(bs, 4 * D, L, device=dev, generator=gen)
    t_bdl = timeit(lambda: selective_scan_fn(u, dl, A, xdbl[:, :, R:R + N], xdbl[:, :, R + N:], Dp, None, bias, True), g)
    # CL: channel-last activations viewed as (B,4D,L)
    ucl = torch.randn(bs, L, 4 * D, device=dev, generator=gen).transpose(1, 2).requires_grad_()
    dcl = (0.5 * torch.rand(bs, L, 4 * D, device=dev, generator=gen)).transpose(1, 2).requires_grad_()
    gcl = torch.randn(bs, L, 4 * D, device=dev, generator=gen).transpose(1, 2)
    t_cl = timeit(lambda: selective_scan_fn(ucl, dcl, A, xdbl[:, :, R:R + N], xdbl[:, :, R + N:], Dp, None, bias, True), gcl)
    # SS2D
    xc = torch.randn(bs, Hh, Hh, D, device=dev, generator=gen).requires_grad_()
    proj = torch.randn(bs, L, 4, C, device=dev, generator=gen).requires_grad_()
    delta = (0.5 * torch.rand(4, bs, L, D, device=dev, generator=gen)).requires_grad_()
    gy = torch.randn(bs, L, D, device=dev, generator=gen)
    t_ss = timeit(lambda: _SS2DScan.apply(xc, proj, delta, None, A, Dp, bias, Hh, Hh, N, R), gy)
    print(f"D={D:4d} L={L:5d}  fwd/bwd ms:  BDL {t_bdl[0]:.3f}/{t_bdl[1]:.3f}   CL {t_cl[0]:.3f}/{t_cl[1]:.3f}   SS2D {t_ss[0]:.3f}/{t_ss[1]:.3f}", flush=True)
