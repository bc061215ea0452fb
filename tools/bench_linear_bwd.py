"""ms_linear_bwd_bf16 (dx and dW from one pass) vs the two ms_gemm_bf16 launches, cold caches, at the stage-0 / stage-1 projection shapes.
usage: python tools/bench_linear_bwd.py [T|B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd import gemm_ops
dev = torch.device("cuda:0")
flush = torch.empty(1 << 28, device=dev, dtype=torch.float32)
def cold(fn, n=7):
    ts = []
    for _ in range(n):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
f32, bf = torch.float32, torch.bfloat16
var = sys.argv[1] if len(sys.argv) > 1 else "T"
cases = {"T": [("in_proj s0", 200704, 192, 48, bf, bf, bf), ("x_proj s0", 200704, 140, 96, f32, f32, f32), ("out_proj s0", 200704, 48, 96, f32, bf, bf)],
         "B": [("in_proj s0", 524288, 256, 64, bf, bf, bf), ("x_proj s0", 524288, 144, 128, f32, f32, f32), ("out_proj s0", 524288, 64, 128, f32, bf, bf)]}[var]
gemm_ops._FUSED_BWD_MIN_ROWS = 1; gemm_ops._FUSED_BWD_MIN_ROW_BYTES = 0
for name, M, N, K, dyt, xt, dxt in cases:
    dy = torch.randn(M, N, device=dev).to(dyt); x = torch.randn(M, K, device=dev).to(xt); w = (torch.randn(N, K, device=dev) * 0.1).to(bf)
    fused = lambda: gemm_ops.linear_bwd_fused(dy, x, w, dxt)
    two = lambda: (gemm_ops.gemm(dy, w, b_trans=True, out_dtype=dxt), gemm_ops.weight_grad(dy, x))
    fused(); two(); torch.cuda.synchronize()
    byts = dy.numel() * dy.element_size() + x.numel() * x.element_size() + M * K * (2 if dxt == bf else 4)
    tf, tt = cold(fused), cold(two)
    print(f"{name:12s} M={M} N={N} K={K}: one pass {tf:7.1f} us ({byts / tf / 1e6:5.2f} TB/s of operand bytes)   two launches {tt:7.1f} us", flush=True)
