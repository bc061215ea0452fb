"""ms_gemm_bf16 vs torch.mm on the SS2D projection shapes with COLD caches (a 1 GiB fill between calls), per tile choice:
what the projections see inside the training step, where 24 ms of other kernels run between two uses of an operand."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd import _lib
from medical_image_classification_amd.gemm_ops import gemm, weight_grad

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
stages = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3]
flush = torch.empty(1 << 28, device=dev, dtype=torch.float32)
def cold(fn, n=6):
    ts = []
    for _ in range(n):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
lib = _lib.lib()
tiles = [(0, 0), (64, 64), (64, 128), (64, 192), (128, 64), (128, 128), (128, 192)]
geo = [(48, 56, 3), (96, 28, 6), (192, 14, 12), (384, 7, 24)]
for si in stages:
    d, Hh, R = geo[si]
    M, D, C4 = bs * Hh * Hh, 2 * d, 4 * (R + 32)
    for name, K, N, a_f32, out_bf16 in (("in_proj", d, 2 * D, False, True), ("x_proj", D, C4, True, False), ("out_proj", D, d, False, True)):
        a = torch.randn(M, K, device=dev); a16 = a.to(torch.bfloat16)
        w = torch.randn(N, K, device=dev) * K ** -0.5; w16 = w.to(torch.bfloat16)
        dy = torch.randn(M, N, device=dev); dy16 = dy.to(torch.bfloat16)
        ain = a if a_f32 else a16
        dyin = dy if name == "x_proj" else dy16
        od = torch.bfloat16 if out_bf16 else torch.float32
        r_f = cold(lambda: torch.mm(a16, w16.t(), out_dtype=torch.float32) if not out_bf16 else torch.mm(a16, w16.t()))
        r_dx = cold(lambda: torch.mm(dy16, w16))
        print(f"stage {si} {name:8s} M={M:6d} K={K:4d} N={N:4d}: torch fwd {r_f:6.1f} dx {r_dx:6.1f}", flush=True)
        for wt, wn in ((w, "f32 W"), (w16, "bf16 W")):
            line = []
            for bm, bn in tiles:
                lib.ms_debug_gemm_tile(bm, bn)
                t_f = cold(lambda: gemm(ain, wt, out_dtype=od))
                t_dx = cold(lambda: gemm(dyin, wt, b_trans=True, out_dtype=torch.float32 if name == "x_proj" else torch.bfloat16))
                line.append(f"{bm}x{bn}: {t_f:5.1f}/{t_dx:5.1f}")
            lib.ms_debug_gemm_tile(0, 0)
            print(f"      {wn:6s} fwd/dx  " + "  ".join(line), flush=True)
        dwbuf = torch.zeros(N, K, device=dev)
        S = 64 if M % 64 == 0 and M // 64 >= 1024 else (16 if M % 16 == 0 and M // 16 >= 1024 else (8 if M // 8 >= 1024 else 2))
        r_dw = cold(lambda: torch.bmm(dy16.view(S, M // S, N).transpose(1, 2), a16.view(S, M // S, K), out_dtype=torch.float32).sum(0))
        line = []
        for bm, bn in tiles:
            lib.ms_debug_gemm_tile(bm, bn)
            line.append(f"{bm}x{bn}: {cold(lambda: weight_grad(dyin, ain, out=dwbuf)):5.1f}")
        lib.ms_debug_gemm_tile(0, 0)
        print(f"      dW torch {r_dw:6.1f}   " + "  ".join(line), flush=True)
