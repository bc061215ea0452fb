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
