"""ms_gemm_bf16 vs torch (hipBLASLt) on the SS2D projection shapes of MedMamba-T bs 64: forward, input gradient, weight
gradient.  us per call and the fraction of the 8 TB/s HBM roofline the call's algorithmic bytes reach."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd.gemm_ops import gemm, weight_grad

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = {"ms": 0.0, "torch": 0.0}
for si, (d, Hh, R, nblk) in enumerate([(48, 56, 3, 2), (96, 28, 6, 2), (192, 14, 12, 4), (384, 7, 24, 2)]):
    M, D, C4 = bs * Hh * Hh, 2 * d, 4 * (R + 32)
    for name, K, N, a_f32, out_bf16 in (("in_proj", d, 2 * D, False, True), ("x_proj", D, C4, True, False), ("out_proj", D, d, False, True)):
        a = torch.randn(M, K, device=dev); a16 = a.to(torch.bfloat16)
        w = torch.randn(N, K, device=dev) * K ** -0.5; w16 = w.to(torch.bfloat16)
        dy = torch.randn(M, N, device=dev); dy16 = dy.to(torch.bfloat16)
        ain = a if a_f32 else a16
        dyin = dy if name == "x_proj" else dy16
        od = torch.bfloat16 if out_bf16 else torch.float32
        t_f = timeit(lambda: gemm(ain, w, out_dtype=od))
        t_dx = timeit(lambda: gemm(dyin, w, b_trans=True, out_dtype=torch.float32 if name == "x_proj" else torch.bfloat16))
        ks = 0
        dwbuf = torch.zeros(N, K, device=dev)
        t_dw = timeit(lambda: weight_grad(dyin, ain, out=dwbuf))
        # torch: bf16 operands prepared outside the timed region (its cast kernels are NOT counted)
        r_f = timeit(lambda: torch.mm(a16, w16.t(), out_dtype=torch.float32) if not out_bf16 else torch.mm(a16, w16.t()))
        r_dx = timeit(lambda: torch.mm(dy16, w16))
        S = 64 if M % 64 == 0 and M // 64 >= 1024 else (16 if M % 16 == 0 and M // 16 >= 1024 else 1)
        r_dw = timeit(lambda: torch.bmm(dy16.view(S, M // S, N).transpose(1, 2), a16.view(S, M // S, K), out_dtype=torch.float32).sum(0))
        esz = lambda f32: 4 if f32 else 2
        b_f = M * K * esz(a_f32) + M * N * (2 if out_bf16 else 4) + N * K * 4
        print(f"stage {si} {name:8s} M={M:6d} K={K:4d} N={N:4d}: fwd {t_f:6.1f} us ({b_f/t_f/8e6*100:4.1f}% HBM) torch {r_f:6.1f} | dx {t_dx:6.1f} torch {r_dx:6.1f} | "
              f"dW {t_dw:6.1f} (splits {ks}) torch {r_dw:6.1f}", flush=True)
        tot["ms"] += nblk * (t_f + t_dx + t_dw); tot["torch"] += nblk * (r_f + r_dx + r_dw)
print(f"per step (10 blocks): ms_gemm_bf16 {tot['ms']/1e3:.3f} ms, torch GEMMs alone {tot['torch']/1e3:.3f} ms (+ its cast / sum kernels)")
