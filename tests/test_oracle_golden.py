"""Pin the CPU oracle (oracle/scan_oracle.c) against vectors produced by RUNNING the reference
(`selective_scan_ref` + torch autograd, selective_scan_interface.py:92-158; SS2D.forward_corev0,
MedMamba.py:386-424) -- see tools/make_golden.py.  CPU only.

Tolerances are the reference's own fp32 ones (test_selective_scan.py:398-401,490-502):
out rtol 6e-4 / atol 2e-3; du 2x; ddelta 5x; dA 1e-3/5e-3 (we use the tighter fwd tol where it holds);
the oracle is a sequential fp32 loop like the reference, so it actually agrees to ~1e-5.
"""
import glob
import os

import numpy as np
import pytest

from oracle import scan_oracle as so

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "scan_*.npz")))


def _close(a, b, rtol, atol, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    assert a.shape == b.shape, what
    assert err.max(initial=-1.0) <= 0, f"{what}: max abs diff {np.abs(a-b).max():.3e} (ref max {np.abs(b).max():.3e})"


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[5:-4] for p in CASES])
def test_scan_oracle_matches_reference(path):
    g = np.load(path)
    get = lambda k: g[k] if k in g.files else None
    sp = bool(int(g["softplus"]))
    out, last = so.scan_fwd(g["u"], g["delta"], g["A"], g["B"], g["C"], get("D"), get("z"), get("delta_bias"), sp)
    _close(out, g["out"], 1e-4, 1e-5, "out")
    _close(last, g["last_state"], 1e-4, 1e-5, "last_state")
    gr = so.scan_bwd(g["u"], g["delta"], g["A"], g["B"], g["C"], get("D"), get("z"), get("delta_bias"), g["g"], sp)
    scale = lambda k: max(1.0, float(np.abs(g[k]).max()))
    for k in ("du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz"):
        if k in g.files:
            _close(gr[k], g[k], 2e-4, 2e-5 * scale(k), k)


@pytest.mark.parametrize("hw", [(3, 5), (4, 4), (7, 2)])
def test_cross_scan_merge_bit_exact(hw, golden_dir):
    H, W = hw
    g = np.load(os.path.join(golden_dir, f"cross_{H}x{W}.npz"))
    x = g["x"].astype(np.float32)
    xs = so.cross_scan(x)
    assert np.array_equal(xs.astype(np.int64), g["xs"])           # permutation: bit exact
    # merge: the spy returned arange(B*4*D*L) as the scan output (values < 2^24: exact in fp32)
    B, _, D, L = g["xs"].shape
    ys = np.arange(B * 4 * D * L, dtype=np.float32).reshape(B, 4, D, L)
    y = so.cross_merge(ys, H, W)
    ref = (g["y1"] + g["y2"] + g["y3"] + g["y4"]).astype(np.float32)
    assert np.array_equal(y, ref)


def test_oracle_torch_wrapper_grad():
    import torch
    g = np.load(CASES[0])
    t = lambda k: torch.from_numpy(g[k]).requires_grad_()
    names = [n for n in ("u", "delta", "A", "B", "C", "D", "delta_bias") if n in g.files]
    ten = {k: t(k) for k in names}
    out = so.selective_scan_oracle(ten["u"], ten["delta"], ten["A"], ten["B"], ten["C"], ten.get("D"), None,
                                   ten.get("delta_bias"), bool(int(g["softplus"])))
    out.backward(torch.from_numpy(g["g"]))
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ten["u"].grad.numpy(), g["du"], rtol=2e-4, atol=2e-5)
