"""Diagnostic for tests/test_train_parity_gpu.py: per-parameter gradient agreement at step 0 of
 (a) this package, bf16 autocast, GPU       vs the fp32 CPU oracle
 (b) this package, fp32, GPU                vs the fp32 CPU oracle
 (c) the CPU oracle under torch.autocast("cpu", bfloat16) (stock autocast of the restatement) vs the fp32 CPU oracle
(c) is the yardstick for what bf16 autocast itself costs at this depth."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from medical_image_classification_amd import medmamba as mm
from oracle import ss2d_oracle

dev = torch.device("cuda:0")
kw = dict(depths=[1, 1, 1, 1], dims=[96, 192, 384, 768], num_classes=8, drop_path_rate=0.0)
torch.manual_seed(0)
net = mm.VSSM(**kw)
sd = {k: v.clone() for k, v in net.state_dict().items()}
x = torch.randn(2, 3, 224, 224); y = torch.tensor([1, 6])
lossf = nn.CrossEntropyLoss()

def cpu_grads(autocast):
    ref = mm.VSSM(**kw); ref.load_state_dict(sd); ss2d_oracle.install(ref); ref.train()
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            loss = lossf(ref(x).float(), y)
    else:
        loss = lossf(ref(x), y)
    loss.backward()
    return float(loss), {k: p.grad.double().flatten() for k, p in ref.named_parameters()}

def gpu_grads(bf16):
    n = mm.VSSM(**kw); n.load_state_dict(sd); n.to(dev).train()
    if bf16:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = lossf(n(x.to(dev)), y.to(dev))
    else:
        loss = lossf(n(x.to(dev)), y.to(dev))
    loss.backward()
    return float(loss), {k: p.grad.float().cpu().double().flatten() for k, p in n.named_parameters()}

l0, g0 = cpu_grads(False)
rows = {}
for name, (l, g) in (("gpu_bf16", gpu_grads(True)), ("gpu_fp32", gpu_grads(False)), ("cpu_autocast", cpu_grads(True))):
    print(f"{name}: loss {l:.6f} vs {l0:.6f}")
    for k in g0:
        r = g0[k]; a = g[k]
        cos = float(a @ r / (a.norm() * r.norm()).clamp_min(1e-300)); rel = float((a - r).norm() / r.norm().clamp_min(1e-300))
        rows.setdefault(k, {})[name] = (cos, rel)
print(f"{'parameter':60s} {'|g|':>9s} | gpu_bf16 cos rel | gpu_fp32 cos rel | cpu_autocast cos rel")
for k, v in rows.items():
    print(f"{k:60s} {float(g0[k].norm()):9.2e} | " + " | ".join(f"{v[n][0]:.5f} {v[n][1]:.4f}" for n in ("gpu_bf16", "gpu_fp32", "cpu_autocast")))
