import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd())
import torch
from medical_image_classification_amd.gemm_ops import gemm
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, f32dy) in [(200704, 192, 48, False), (200704, 140, 96, True), (200704, 48, 96, False), (50176, 384, 96, False), (12544, 768, 192, False), (3136, 1536, 384, False)]:
    dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    if not f32dy: dy = dy.to(torch.bfloat16)
    out = torch.zeros(N, K, device=dev); outT = torch.zeros(K, N, device=dev)
    res = []
    for S in (8, 16, 32, 64, 128, 256, 512):
        if M // S < 64: continue
        t1 = timeit(lambda: gemm(dy, x, a_trans=True, b_trans=True, out=out, accumulate=True, k_splits=S))
        t2 = timeit(lambda: gemm(x, dy, a_trans=True, b_trans=True, out=outT, accumulate=True, k_splits=S))
        res.append(f"S={S}: {t1:.0f}/{t2:.0f}")
    print(f"M={M} N={N} K={K}  dW / dW^T us: " + "  ".join(res), flush=True)
