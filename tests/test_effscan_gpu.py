"""cross_selective_scan / cross_selective_scan_cross (FusionMamba cross.py:193-414) on the HIP scan kernels
against vectors produced by running the reference functions on CPU with the reference's own selective_scan_ref behind them
(tools/make_golden_cross.py): output and every gradient.  Tolerances: forward 1e-3 relative (north-star), gradients the
reference test's fp32 rows scaled to the tensor's magnitude."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
dev = lambda: torch.device("cuda:0")


def close(got, want, tol, msg):
    np.testing.assert_allclose(got.detach().float().cpu().numpy(), want, rtol=tol, atol=max(1e-6, tol * float(np.abs(want).max())),
                               err_msg=msg)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "effscan_core_*.npz"))), ids=os.path.basename)
@pytest.mark.parametrize("mode", ["single", "cross"])
def test_cross_selective_scan_matches_reference_vectors(path, mode):
    from medical_image_classification_amd import efficient_scan as es
    g = np.load(path)
    t = lambda k: torch.from_numpy(g[k]).to(dev()).requires_grad_()
    names = ("x_proj_weight", "x_proj_bias", "dt_projs_weight", "dt_projs_bias", "A_logs", "Ds")
    p = {k: t(k) for k in names}
    x1, x2 = t("x1"), t("x2")
    D = g["x1"].shape[1]
    norm = torch.nn.LayerNorm(D).to(dev())
    with torch.no_grad():
        norm.weight.copy_(torch.from_numpy(g["norm_w"])); norm.bias.copy_(torch.from_numpy(g["norm_b"]))
    if mode == "single":
        y = es.cross_selective_scan(x1, *[p[k] for k in names], out_norm=norm, nrows=1)
        assert es.cross_selective_scan_new is es.cross_selective_scan
    else:
        y = es.cross_selective_scan_cross(x1, x2, *[p[k] for k in names], out_norm=norm, nrows=-1)
    y.backward(torch.from_numpy(g[f"{mode}_gy"]).to(dev()))
    close(y, g[f"{mode}_y"], 1e-3, "y")
    close(x1.grad, g[f"{mode}_dx1"], 2e-3, "dx1")
    if mode == "cross":
        close(x2.grad, g[f"{mode}_dx2"], 2e-3, "dx2")
    for k in names:
        close(p[k].grad, g[f"{mode}_d{k}"], 5e-3, f"d{k}")
    close(norm.weight.grad, g[f"{mode}_dnorm_w"], 2e-3, "dnorm_w")


def test_selective_scan_namespace_and_gpu_permutations():
    from medical_image_classification_amd import efficient_scan as es
    x = torch.randn(2, 5, 11, 8, device=dev())
    xs = es.EfficientScan.apply(x, 2)
    assert xs.shape == (2, 4, 5, 6 * 4)
    assert torch.equal(es.EfficientMerge.apply(xs, 11, 8, 2).view_as(x), x)
    assert torch.equal(xs.cpu(), es.EfficientScan.apply(x.cpu(), 2))
    with pytest.raises(AssertionError):
        es.SelectiveScan.apply(x, x, x, x, x, None, None, True, 8)


@pytest.mark.parametrize("cfg", [(2, 96, 28, 28, 16, 6), (1, 48, 9, 14, 16, 3), (2, 32, 64, 64, 8, 2)])
def test_lattice_addressing_mode_matches_gather_formulation(cfg, monkeypatch):
    """The stride-2 sub-lattices as an addressing mode of the scan kernels (MS_SCAN_LATTICE: no sequences materialised) against
    the gather formulation of the same functions (EfficientScan -> selective_scan_fn -> EfficientMerge, itself pinned by the
    reference-run vectors above) at FusionMamba-sized maps: output and every gradient."""
    from medical_image_classification_amd import efficient_scan as es
    B, D, H, W, N, R = cfg
    gen = torch.Generator().manual_seed(11)
    C = R + 2 * N
    mk = lambda *s, scale=1.0: (torch.randn(*s, generator=gen) * scale).to(dev())
    base = dict(x=mk(B, D, H, W), xw=mk(4, C, D, scale=D ** -0.5), dw=mk(4, D, R, scale=R ** -0.5), db=mk(4, D, scale=0.5) - 3.0,
                al=torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(4 * D, 1).to(dev()), ds=torch.ones(4 * D, device=dev()))
    gy = mk(B, H, W, D)
    res = []
    for native in (True, False):
        monkeypatch.setattr(es, "_LATTICE_KERNEL", native)
        t = {k: v.clone().requires_grad_() for k, v in base.items()}
        norm = torch.nn.LayerNorm(D).to(dev())
        y = es.cross_selective_scan(t["x"], t["xw"], None, t["dw"], t["db"], t["al"], t["ds"], out_norm=norm, nrows=1)
        y.backward(gy)
        res.append((y.detach(), {k: v.grad for k, v in t.items()}, norm.weight.grad))
    (y1, g1, n1), (y0, g0, n0) = res
    close(y1, y0.cpu().numpy(), 1e-4, "y")
    for k in base:
        close(g1[k], g0[k].cpu().numpy(), 2e-3, f"d{k}")
    close(n1, n0.cpu().numpy(), 2e-3, "dnorm_w")
