"""Host-side (enqueue) and device cost of the batched GEMMs used by the projections, per call."""
import time, torch
dev = torch.device("cuda:0")
M, D, R = 64 * 56 * 56, 96, 3
dts = torch.randn(4, M, R, device=dev); W = torch.randn(4, D, R, device=dev); dd = torch.randn(4, M, D, device=dev)
S = 64
xb = torch.randn(M, 48, device=dev, dtype=torch.bfloat16); dyb = torch.randn(M, 192, device=dev, dtype=torch.bfloat16)
cases = {
    "dt fwd  bmm (4,M,3)x(4,3,96) fp32": lambda: torch.bmm(dts, W.transpose(1, 2)),
    "dt ddts bmm (4,M,96)x(4,96,3) fp32": lambda: torch.bmm(dd, W),
    "dt dW   bmm (256,96,M/64)x(256,M/64,3) fp32": lambda: torch.bmm(dd.view(4 * S, M // S, D).transpose(1, 2), dts.view(4 * S, M // S, R)),
    "dt fwd  4x mm": lambda: [torch.mm(dts[k], W[k].t()) for k in range(4)],
    "dt ddts 4x mm": lambda: [torch.mm(dd[k], W[k]) for k in range(4)],
    "dt dW   einsum": lambda: torch.einsum("kmd,kmr->kdr", dd, dts),
    "lin dW  bmm (128,192,M/128)x(128,M/128,48) bf16": lambda: torch.bmm(dyb.view(128, M // 128, 192).transpose(1, 2), xb.view(128, M // 128, 48)),
    "lin fwd mm (M,48)x(48,192) bf16": lambda: torch.mm(xb, torch.randn(48, 192, device=dev, dtype=torch.bfloat16)),
}
for name, fn in cases.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:52s} host {1e6 * (t1 - t0) / n:8.1f} us/call   total {1e6 * (t2 - t0) / n:8.1f} us/call", flush=True)
