"""x_proj weight gradient at stage 0 (dW (140 x 96) = dproj(200704 x 140 fp32)^T @ xc(200704 x 96 fp32)): tile / orientation / split sweep
of ms_gemm_bf16's accumulating modes, cold cache (1 GiB fill between calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd import _lib
from medical_image_classification_amd.gemm_ops import gemm
dev = torch.device("cuda:0"); lib = _lib.lib()
flush = torch.empty(1 << 28, device=dev, dtype=torch.float32)
def cold(fn, n=5):
    ts = []
    for _ in range(n):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for (M, N, K) in [(200704, 140, 96), (50176, 152, 192)]:
    dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
    ref = (dy.to(torch.bfloat16).float().t() @ x.to(torch.bfloat16).float())
    out = torch.zeros(N, K, device=dev); outT = torch.zeros(N, K, device=dev)
    for (bm, bn) in [(0, 0), (64, 64), (64, 128), (128, 64), (128, 128), (64, 192), (128, 192)]:
        lib.ms_debug_gemm_tile(bm, bn)
        res = []
        for S in (16, 32, 64, 128, 256):
            out.zero_(); gemm(dy, x, a_trans=True, b_trans=True, out=out, accumulate=True, k_splits=S)
            e1 = float((out - ref).abs().max() / ref.abs().max())
            t1 = cold(lambda: gemm(dy, x, a_trans=True, b_trans=True, out=out, accumulate=True, k_splits=S))
            def tr():
                _lib.check(lib.ms_gemm_bf16(x.data_ptr(), 1, 1, x.stride(0), dy.data_ptr(), 1, 1, dy.stride(0), outT.data_ptr(), 3, outT.stride(0), K, N, M, S,
                                            _lib.current_stream_ptr(dev)), "gemm")
            outT.zero_(); tr()
            e2 = float((outT - ref).abs().max() / ref.abs().max())
            t2 = cold(tr)
            res.append(f"S={S}: {t1:.0f}/{t2:.0f}" + ("" if max(e1, e2) < 1e-2 else f" ERR {e1:.1e} {e2:.1e}"))
        print(f"M={M} N={N} K={K} tile {bm}x{bn}  (N rows / K rows) us: " + "  ".join(res), flush=True)
lib.ms_debug_gemm_tile(0, 0)
