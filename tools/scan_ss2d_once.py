"""One SS2D-mode scan problem (MedMamba-T stage shape, fwd + bwd) a few times: target for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd.ss2d_fused import _SS2DScan

dev = torch.device("cuda:0")
D, Hh, R = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (192, 56, 6)))
bs = int(sys.argv[4]) if len(sys.argv) > 4 else 64
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
L, N, C = Hh * Hh, 16, R + 32
gen = torch.Generator(device=dev).manual_seed(0)
A = (-0.5 * torch.rand(4 * D, N, device=dev, generator=gen)).requires_grad_()
Dp = torch.randn(4 * D, device=dev, generator=gen).requires_grad_()
bias = (0.5 * torch.rand(4 * D, device=dev, generator=gen)).requires_grad_()
xc = torch.randn(bs, Hh, Hh, D, device=dev, generator=gen).requires_grad_()
proj = torch.randn(bs, L, 4, C, device=dev, generator=gen).requires_grad_()
delta = (0.5 * torch.rand(4, bs, L, D, device=dev, generator=gen)).requires_grad_()
gy = torch.randn(bs, L, D, device=dev, generator=gen)
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for i in range(iters):
    e[0].record(); out = _SS2DScan.apply(xc, proj, delta, None, A, Dp, bias, Hh, Hh, N, R); e[1].record()
    out.backward(gy); e[2].record(); torch.cuda.synchronize()
    print(f"iter {i}: fwd {e[0].elapsed_time(e[1]):.3f} ms  bwd {e[1].elapsed_time(e[2]):.3f} ms", flush=True)
