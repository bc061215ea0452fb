"""FusionMamba's module classes (medical_image_classification_amd/cross.py <- CrossMamba/FusionMamba/models/cross.py:417-1384,
SURVEY.md 8f-3) on the HIP scan kernels (stride-2 sub-lattice addressing mode) against vectors produced by RUNNING the reference
classes on CPU with the reference's own selective_scan_ref behind them (tools/make_golden_crossmods.py): the reference
state_dict loads strictly, outputs agree to 1e-3 relative (north-star), input and parameter gradients to the fp32 rows of the
reference's own operator test scaled to each tensor's magnitude."""
import glob
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
dev = lambda: torch.device("cuda:0")

BUILD = {
    "ss2d_d12_5x6": lambda m: m.SS2D(d_model=12, d_state=4),
    "ss2d_nozact_lowrank_8x8": lambda m: m.SS2D(d_model=16, d_state=3, ssm_ratio=2.0, ssm_rank_ratio=1.0, forward_type="v2nozact"),
    "ss2d_cross_d12_6x7": lambda m: m.SS2D_cross_new(d_model=12, d_state=4),
    "vssblock_new_d16_6x6": lambda m: m.VSSBlock_new(hidden_dim=16, ssm_d_state=4, mlp_ratio=2.0),
    "vssblock_cross_d16_5x8": lambda m: m.VSSBlock_Cross_new(hidden_dim=16, d_state=4),
}


def _load(tag):
    g = np.load(os.path.join(GOLD, f"crossmods_{tag}.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    return g, sd


@pytest.mark.parametrize("tag", sorted(BUILD))
def test_reference_state_dict_loads_strictly(tag):
    """CPU: same parameter / buffer names and shapes as the reference classes (what a checkpoint written by the reference needs)."""
    from medical_image_classification_amd import cross
    g, sd = _load(tag)
    mod = BUILD[tag](cross)
    assert set(mod.state_dict().keys()) == set(sd.keys())
    mod.load_state_dict(sd, strict=True)
    for k, v in mod.state_dict().items():
        assert v.shape == sd[k].shape, k
    if tag == "ss2d_d12_5x6":
        # the other cores of cross.py:475-485: v0 / v0_seq are built (full-resolution scan); share_ssm / share_a are `...` in the
        # reference -- their parameter shapes (K = 1 / K2 = 1) are the reference's, calling them raises
        assert cross.SS2D(d_model=12, forward_type="v0").forward_core.__name__ == "forward_corev0"
        ssm = cross.SS2D(d_model=12, d_state=4, forward_type="share_ssm")
        assert ssm.x_proj_weight.shape[0] == 1 and ssm.A_logs.shape[0] == 24
        sa = cross.SS2D(d_model=12, d_state=4, forward_type="share_a")
        assert sa.x_proj_weight.shape[0] == 4 and sa.A_logs.shape[0] == 24 and sa.Ds.shape[0] == 24
        with pytest.raises(NotImplementedError):
            ssm.forward_core(torch.zeros(1, 2, 2, 24))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(BUILD))
def test_module_matches_reference_vectors(tag):
    from medical_image_classification_amd import cross
    g, sd = _load(tag)
    mod = BUILD[tag](cross)
    mod.load_state_dict(sd, strict=True)
    mod = mod.to(dev()).train()
    xs = [torch.from_numpy(g[k]).to(dev()).requires_grad_() for k in sorted(k for k in g.files if k[0] == "x" and k[1:].isdigit())]
    y = mod(*xs)
    y.backward(torch.from_numpy(g["gy"]).to(dev()))

    def close(got, want, tol, msg):
        np.testing.assert_allclose(got.detach().float().cpu().numpy(), want, rtol=tol, atol=max(1e-6, tol * float(np.abs(want).max())),
                                   err_msg=msg)
    close(y, g["y"], 1e-3, "y")
    for i, x in enumerate(xs):
        close(x.grad, g[f"dx{i}"], 2e-3, f"dx{i}")
    n = 0
    for name, p in mod.named_parameters():
        key = f"grad.{name}"
        if key in g.files:
            assert p.grad is not None, name
            close(p.grad, g[key], 5e-3, key); n += 1
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name        # unused by the reference as well
    assert n == sum(1 for k in g.files if k.startswith("grad."))


CORE = {"v0_d12_5x6": dict(d_model=12, d_state=4, forward_type="v0"), "v0_d16_7x4": dict(d_model=16, d_state=5, forward_type="v0")}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(CORE))
def test_v0_core_matches_reference_vectors(tag):
    """SS2D.forward_corev0 of cross.py:598-646 (the full-resolution four-direction scan + out_norm) CALLED DIRECTLY, as the reference
    allows (its forward() cannot reach the v0 cores: it passes `step_size=`, cross.py:732): output, input gradient and the seven
    parameter gradients against vectors from running the reference's method on CPU (tools/make_golden_crossmods.py dump_core).
    forward_corev0_seq is the same function here (the reference's cannot run: its wrapper misplaces delta_bias, cross.py:650)."""
    from medical_image_classification_amd import cross
    g = np.load(os.path.join(GOLD, f"crosscore_{tag}.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    mod = cross.SS2D(**CORE[tag])
    assert set(mod.state_dict().keys()) == set(sd.keys())
    mod.load_state_dict(sd, strict=True)
    mod = mod.to(dev()).train()
    x = torch.from_numpy(g["x0"]).to(dev()).requires_grad_()
    y = mod.forward_corev0(x)
    y.backward(torch.from_numpy(g["gy"]).to(dev()))

    def close(got, want, tol, msg):
        np.testing.assert_allclose(got.detach().float().cpu().numpy(), want, rtol=tol, atol=max(1e-6, tol * float(np.abs(want).max())),
                                   err_msg=msg)
    close(y, g["y"], 1e-3, "y")
    close(x.grad, g["dx0"], 2e-3, "dx")
    n = 0
    for name, p in mod.named_parameters():
        if f"grad.{name}" in g.files:
            close(p.grad, g[f"grad.{name}"], 5e-3, name); n += 1
    assert n == 7
    with torch.no_grad():                     # v0_seq and forward() through the v0 core: same function
        assert torch.equal(mod.forward_corev0_seq(x.detach()), mod.forward_corev0(x.detach()))
        out = mod(torch.randn(1, 5, 6, CORE[tag]["d_model"], device=dev()))
        assert out.shape[-1] == CORE[tag]["d_model"] and torch.isfinite(out).all()
