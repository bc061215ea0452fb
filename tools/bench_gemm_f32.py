"""ms_gemm_f32 against torch's fp32 GEMMs (hipBLASLt) at the MedMamba-T projection shapes, bs 64: forward / dx / dW, warm caches.
usage: python tools/bench_gemm_f32.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd.gemm_ops import gemm_f32, weight_grad_f32

dev = torch.device("cuda:0")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
bs = 64
for st, (dm, hw, R) in enumerate([(48, 56, 3), (96, 28, 6), (192, 14, 12), (384, 7, 24)]):
    M, D = bs * hw * hw, 2 * dm
    for name, N, K in (("in_proj", 2 * D, dm), ("x_proj", 4 * (R + 32), D), ("out_proj", dm, D)):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); dy = torch.randn(M, N, device=dev)
        f = t(lambda: gemm_f32(x, w)); fl = t(lambda: torch.mm(x, w.t()))
        b = t(lambda: gemm_f32(dy, w.t().contiguous())); bl = t(lambda: torch.mm(dy, w))
        g = t(lambda: weight_grad_f32(dy, x)); gl = t(lambda: torch.mm(dy.t(), x))
        print(f"stage {st} {name:8s} M={M:6d} N={N:4d} K={K:4d}: fwd {f:6.1f} vs {fl:6.1f} | dx {b:6.1f} vs {bl:6.1f} | dW {g:6.1f} vs {gl:6.1f} us", flush=True)
