"""Golden vectors for medical_image_classification_amd/cross.py, produced by RUNNING the reference's module classes
(CrossMamba/FusionMamba/models/cross.py: SS2D, SS2D_cross_new, VSSBlock_new, VSSBlock_Cross_new) on CPU in this container
(SURVEY.md 8f-3): state_dict, inputs, outputs, input gradients and every parameter gradient.

Environment stubs, as in tools/make_golden_cross.py: the absent packages, `selective_scan_cuda.fwd / bwd` served by the
reference's OWN pure-torch `selective_scan_ref`, and -- because cross.py calls `.cuda()` inside two constructors (:800, :826) --
`torch.Tensor.cuda` as a no-op while the modules are built.  Nothing is copied: the vectors are what the reference code computes.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_crossmods.py      # writes tests/golden/crossmods_*.npz
"""
import sys
sys.dont_write_bytecode = True
import os, types
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg                                                        # noqa: E402  (stubs + loader)

ssi, _ = mg.import_reference()
ext = sys.modules["selective_scan_cuda"]


def _fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus):
    # x: the (batch, dim, n_chunks, 2 * dstate) checkpoint tensor of the CUDA extension; SelectiveScanFn.forward slices it for last_state
    return [ssi.selective_scan_ref(u, delta, A, B, C, D, z, delta_bias, delta_softplus), torch.zeros(u.shape[0], u.shape[1], 1, 2 * A.shape[1])]


def _bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, out, dz, delta_softplus, recompute):
    ins = [t.detach().clone().requires_grad_() if t is not None else None for t in (u, delta, A, B, C, D, delta_bias)]
    with torch.enable_grad():
        ssi.selective_scan_ref(ins[0], ins[1], ins[2], ins[3], ins[4], ins[5], None, ins[6], delta_softplus).backward(dout)
    return [t.grad if t is not None else None for t in ins]


ext.fwd, ext.bwd = _fwd, _bwd
pkg = types.ModuleType("mamba_ssm"); pkg.Mamba = object
sys.modules.update({"mamba_ssm": pkg, "mamba_ssm.ops": types.ModuleType("mamba_ssm.ops"), "mamba_ssm.ops.selective_scan_interface": ssi})
torch.Tensor.cuda = lambda self, *a, **k: self
cross = mg._load_file("ref_cross", f"{mg.REF}/CrossMamba/FusionMamba/models/cross.py")


def dump(tag, mod, inputs):
    # move every parameter off its symmetric initial value so that no gradient is trivially zero
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if p.requires_grad and ("mask" in n or "theta" in n or n.endswith("Ds") or "norm" in n or "ln_" in n):
                p.add_(torch.randn_like(p) * 0.2)
    xs = [x.clone().requires_grad_() for x in inputs]
    y = mod(*xs)
    gy = torch.randn_like(y)
    y.backward(gy)
    out = {f"sd.{k}": v.detach().numpy() for k, v in mod.state_dict().items()}
    out.update({f"x{i}": x.detach().numpy() for i, x in enumerate(xs)})
    out.update({f"dx{i}": x.grad.numpy() for i, x in enumerate(xs)})
    out.update(y=y.detach().numpy(), gy=gy.numpy())
    out.update({f"grad.{n}": p.grad.numpy() for n, p in mod.named_parameters() if p.grad is not None})
    np.savez_compressed(os.path.join(mg.OUT, f"crossmods_{tag}.npz"), **{k: np.asarray(v) for k, v in out.items()})
    print(tag, "y", tuple(y.shape), "params with grad", sum(1 for p in mod.parameters() if p.grad is not None))


torch.manual_seed(0)
dump("ss2d_d12_5x6", cross.SS2D(d_model=12, d_state=4), [torch.randn(2, 5, 6, 12)])
dump("ss2d_nozact_lowrank_8x8", cross.SS2D(d_model=16, d_state=3, ssm_ratio=2.0, ssm_rank_ratio=1.0, forward_type="v2nozact"), [torch.randn(1, 8, 8, 16)])
dump("ss2d_cross_d12_6x7", cross.SS2D_cross_new(d_model=12, d_state=4), [torch.randn(2, 6, 7, 12), torch.randn(2, 6, 7, 12)])
dump("vssblock_new_d16_6x6", cross.VSSBlock_new(hidden_dim=16, ssm_d_state=4, mlp_ratio=2.0), [torch.randn(2, 6, 6, 16)])
dump("vssblock_cross_d16_5x8", cross.VSSBlock_Cross_new(hidden_dim=16, d_state=4), [torch.randn(1, 5, 8, 16), torch.randn(1, 5, 8, 16)])


def dump_core(tag, mod, x):
    """forward_corev0 CALLED DIRECTLY (the reference's forward() cannot reach it: it passes `step_size=`, which the v0 signatures do
    not take, cross.py:732 vs :598): y, dx and the core's parameter gradients.  forward_corev0_seq cannot run at all in the reference:
    its local wrapper hands (.., D, delta_bias, delta_softplus) positionally to selective_scan_fn, whose seventh parameter is `z`
    (cross.py:650 vs selective_scan_interface.py:83) -- a TypeError here, a shape error under the CUDA extension."""
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if n.endswith("Ds") or "norm" in n:
                p.add_(torch.randn_like(p) * 0.2)
    xi = x.clone().requires_grad_()
    y = mod.forward_corev0(xi)
    gy = torch.randn_like(y)
    y.backward(gy)
    out = {f"sd.{k}": v.detach().numpy() for k, v in mod.state_dict().items()}
    out.update(x0=x.numpy(), dx0=xi.grad.numpy(), y=y.detach().numpy(), gy=gy.numpy())
    out.update({f"grad.{n}": p.grad.numpy() for n, p in mod.named_parameters() if p.grad is not None})
    np.savez_compressed(os.path.join(mg.OUT, f"crosscore_{tag}.npz"), **{k: np.asarray(v) for k, v in out.items()})
    print(tag, "y", tuple(y.shape), "params with grad", sum(1 for p in mod.parameters() if p.grad is not None))


dump_core("v0_d12_5x6", cross.SS2D(d_model=12, d_state=4, forward_type="v0"), torch.randn(2, 5, 6, 24))
dump_core("v0_d16_7x4", cross.SS2D(d_model=16, d_state=5, forward_type="v0"), torch.randn(1, 7, 4, 32))
