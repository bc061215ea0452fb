"""csrc/ssd_chunk.hip against a float64 sequential evaluation of the SSD recurrence on the GPU: output and all seven gradients."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from medical_image_classification_amd import cnn_mamba as cm
dev = torch.device("cuda:0")
def ref64(x, dt, A, B, C, D, bias):
    b, l, h, p = x.shape; n = B.shape[3]
    dtv = F.softplus(dt + bias)
    state = torch.zeros(b, h, p, n, dtype=torch.float64, device=dev)
    ys = []
    for t in range(l):
        a = torch.exp(dtv[:, t] * A)
        state = a[:, :, None, None] * state + (dtv[:, t, :, None] * x[:, t])[..., None] * B[:, t, 0][:, None, None, :]
        ys.append((state * C[:, t, 0][:, None, None, :]).sum(-1))
    return torch.stack(ys, 1) + x * (D if D.dim() == 2 else D[:, None])
ok = True
for (b, l, h, p, n, hd) in [(2, 200, 4, 64, 64, False), (1, 64, 2, 64, 128, True), (2, 49, 8, 64, 512, False), (1, 300, 3, 64, 64, False)]:
    g = torch.Generator(device=dev).manual_seed(l + n)
    mk = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
    base = dict(x=mk(b, l, h, p), dt=mk(b, l, h) - 1.0, A=-(torch.rand(h, device=dev, generator=g) * 4 + 0.2), B=mk(b, l, 1, n, sc=0.3),
                C=mk(b, l, 1, n, sc=0.3), D=mk(h, p) if hd else mk(h), bias=mk(h, sc=0.5))
    gy = mk(b, l, h, p)
    t64 = {k: v.double().requires_grad_() for k, v in base.items()}
    want = ref64(*[t64[k] for k in ("x", "dt", "A", "B", "C", "D", "bias")])
    want.backward(gy.double())
    t32 = {k: v.clone().requires_grad_() for k, v in base.items()}
    got = cm._SSDChunkKernels.apply(t32["x"], t32["dt"], t32["A"], t32["B"], t32["C"], t32["D"], t32["bias"], True)
    got.backward(gy)
    torch.cuda.synchronize()
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max().clamp_min(1e-30))
    line = f"b{b} l{l} h{h} n{n}: y {rel(got, want):.1e}"
    for k in base:
        e = rel(t32[k].grad, t64[k].grad); line += f"  d{k} {e:.1e}"; ok = ok and e < 2e-4
    print(line, flush=True)
print("OK" if ok else "MISMATCH")
