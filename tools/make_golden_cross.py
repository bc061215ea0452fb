"""Golden vectors for medical_image_classification_amd/efficient_scan.py, produced by RUNNING the reference's
CrossMamba/FusionMamba/models/cross.py on CPU in this container: `EfficientScan`, `EfficientMerge`
(forward and backward) and `cross_selective_scan` / `cross_selective_scan_cross` (forward and all gradients).

cross.py hard-imports packages that are absent here; they are stubbed exactly as tools/make_golden.py does, plus:
  * `selective_scan_cuda` (the CUDA extension cross.py's `SelectiveScan` calls, cross.py:119,130): `fwd` / `bwd` are served by
    the reference's OWN pure-torch `selective_scan_ref` (mamba_ssm/ops/selective_scan_interface.py:92-158) and its autograd.
Nothing is copied: the vectors are what the reference code computes on seeded inputs.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_cross.py      # writes tests/golden/effscan_*.npz
"""
import sys
sys.dont_write_bytecode = True
import os, types
import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg                                                        # noqa: E402  (stubs + loader)

ssi, _ = mg.import_reference()
ext = sys.modules["selective_scan_cuda"]


def _fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus):
    out = ssi.selective_scan_ref(u, delta, A, B, C, D, z, delta_bias, delta_softplus)
    return [out, torch.zeros(1)]


def _bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, out, dz, delta_softplus, recompute):
    ins = [t.detach().clone().requires_grad_() if t is not None else None for t in (u, delta, A, B, C, D, delta_bias)]
    with torch.enable_grad():
        o = ssi.selective_scan_ref(ins[0], ins[1], ins[2], ins[3], ins[4], ins[5], None, ins[6], delta_softplus)
        o.backward(dout)
    return [t.grad if t is not None else None for t in ins]


ext.fwd, ext.bwd = _fwd, _bwd
pkg = types.ModuleType("mamba_ssm"); pkg.Mamba = object
ops = types.ModuleType("mamba_ssm.ops")
sys.modules.update({"mamba_ssm": pkg, "mamba_ssm.ops": ops, "mamba_ssm.ops.selective_scan_interface": ssi})
cross = mg._load_file("ref_cross", f"{mg.REF}/CrossMamba/FusionMamba/models/cross.py")

torch.manual_seed(0)
for tag, (B, C, H, W) in {"5x6": (2, 4, 5, 6), "4x4": (1, 6, 4, 4), "7x3": (1, 4, 7, 3), "1x1": (1, 2, 1, 1)}.items():
    out = {}
    x = torch.randn(B, C, H, W, requires_grad=True)
    xs = cross.EfficientScan.apply(x, 2)
    g = torch.randn_like(xs)
    xs.backward(g)
    out.update(x=x.detach().numpy(), xs=xs.detach().numpy(), g_xs=g.numpy(), dx=x.grad.numpy())
    ys = torch.randn(B, 4, C, xs.shape[-1], requires_grad=True)
    y = cross.EfficientMerge.apply(ys, H, W, 2)
    gy = torch.randn_like(y)
    y.backward(gy)
    out.update(ys=ys.detach().numpy(), y=y.detach().numpy(), g_y=gy.numpy(), dys=ys.grad.numpy())
    np.savez_compressed(os.path.join(mg.OUT, f"effscan_perm_{tag}.npz"), **out)

for tag, (B, D, H, W, N, R) in {"d8_5x6": (2, 8, 5, 6, 4, 2), "d6_7x4": (1, 6, 7, 4, 3, 1)}.items():
    K = 4
    mk = lambda *s, sc=1.0: (torch.randn(*s) * sc).requires_grad_()
    x1, x2 = mk(B, D, H, W), mk(B, D, H, W)
    p = dict(x_proj_weight=mk(K, R + 2 * N, D, sc=0.3), x_proj_bias=mk(K, R + 2 * N, sc=0.1), dt_projs_weight=mk(K, D, R, sc=0.5),
             dt_projs_bias=mk(K, D, sc=0.5), A_logs=torch.log(torch.arange(1, N + 1).float()).repeat(K * D, 1).requires_grad_(),
             Ds=mk(K * D))
    norm = nn.LayerNorm(D)
    with torch.no_grad():
        norm.weight.add_(torch.randn(D) * 0.2); norm.bias.add_(torch.randn(D) * 0.2)
    out = {k: v.detach().numpy() for k, v in p.items()}
    out.update(x1=x1.detach().numpy(), x2=x2.detach().numpy(), norm_w=norm.weight.detach().numpy(), norm_b=norm.bias.detach().numpy())
    for name, fn, args in (("single", cross.cross_selective_scan, (x1,)), ("cross", cross.cross_selective_scan_cross, (x1, x2))):
        for t in list(p.values()) + [x1, x2, norm.weight, norm.bias]:
            t.grad = None
        y = fn(*args, p["x_proj_weight"], p["x_proj_bias"], p["dt_projs_weight"], p["dt_projs_bias"], p["A_logs"], p["Ds"],
               out_norm=norm, nrows=1, delta_softplus=True, to_dtype=True, step_size=2)
        gy = torch.randn_like(y)
        y.backward(gy)
        out.update({f"{name}_y": y.detach().numpy(), f"{name}_gy": gy.numpy(), f"{name}_dx1": x1.grad.numpy(),
                    f"{name}_dnorm_w": norm.weight.grad.numpy()})
        if name == "cross":
            out[f"{name}_dx2"] = x2.grad.numpy()
        out.update({f"{name}_d{k}": v.grad.numpy() for k, v in p.items()})
    np.savez_compressed(os.path.join(mg.OUT, f"effscan_core_{tag}.npz"), **{k: np.asarray(v, dtype=np.float32) for k, v in out.items()})
print("wrote", sorted(f for f in os.listdir(mg.OUT) if f.startswith("effscan")))
