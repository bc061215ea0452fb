"""oracle/scan_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy/ctypes front-end of oracle/scan_oracle.c (CPU restatement of the reference's
`selective_scan_ref`, selective_scan_interface.py:92-158, and of the adjoint the CUDA kernel
implements, selective_scan_bwd_kernel.cuh:140-475), plus a torch.autograd wrapper so the oracle can be
dropped into the module surface for CPU model-level checks.

Parity pin: tests/test_oracle_golden.py checks every function here against tests/golden/scan_*.npz and
cross_*.npz, which were produced by running the reference itself (tools/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libscan_oracle.so")
    src = os.path.join(_HERE, "scan_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        fp, ci = ctypes.c_void_p, ctypes.c_int
        _LIB.ms_oracle_scan_fwd.argtypes = [fp] * 8 + [ci] * 6 + [fp, fp]
        _LIB.ms_oracle_scan_fwd.restype = ci
        _LIB.ms_oracle_scan_bwd.argtypes = [fp] * 9 + [ci] * 6 + [fp] * 8
        _LIB.ms_oracle_scan_bwd.restype = ci
        _LIB.ms_oracle_cross_scan.argtypes = [fp, ci, ci, ci, ci, fp]
        _LIB.ms_oracle_cross_merge.argtypes = [fp, ci, ci, ci, ci, fp]
    return _LIB


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _norm_bc(B, C):
    """(B,N,L) -> (B,1,N,L) like SelectiveScanFn.forward (selective_scan_interface.py:37-42)."""
    sq_b = B.ndim == 3
    sq_c = C.ndim == 3
    if sq_b:
        B = B[:, None]
    if sq_c:
        C = C[:, None]
    return B, C, sq_b, sq_c


def scan_fwd(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False):
    """Returns (out, last_state).  Shapes as selective_scan_ref (selective_scan_interface.py:94-105)."""
    u, delta, A, D, z, delta_bias = map(_c, (u, delta, A, D, z, delta_bias))
    B, C, _, _ = _norm_bc(_c(B), _c(C))
    B, C = _c(B), _c(C)
    batch, dim, L = u.shape
    N, G = A.shape[1], B.shape[1]
    assert delta.shape == u.shape and A.shape[0] == dim and B.shape == (batch, G, N, L) and C.shape == B.shape
    out = np.empty_like(u)
    last = np.empty((batch, dim, N), np.float32)
    rc = lib().ms_oracle_scan_fwd(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(z), _p(delta_bias),
                                  int(bool(delta_softplus)), batch, dim, N, L, G, _p(out), _p(last))
    if rc:
        raise RuntimeError(f"ms_oracle_scan_fwd failed rc={rc}")
    return out, last


def scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, delta_softplus=False):
    """Returns dict(du, ddelta, dA, dB, dC, dD, ddelta_bias, dz)."""
    u, delta, A, D, z, delta_bias, dout = map(_c, (u, delta, A, D, z, delta_bias, dout))
    B, C, sq_b, sq_c = _norm_bc(_c(B), _c(C))
    B, C = _c(B), _c(C)
    batch, dim, L = u.shape
    N, G = A.shape[1], B.shape[1]
    du, dd = np.empty_like(u), np.empty_like(u)
    dA, dB, dC = np.empty_like(A), np.empty_like(B), np.empty_like(C)
    dD = np.empty(dim, np.float32) if D is not None else None
    db = np.empty(dim, np.float32) if delta_bias is not None else None
    dz = np.empty_like(u) if z is not None else None
    rc = lib().ms_oracle_scan_bwd(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(z), _p(delta_bias), _p(dout),
                                  int(bool(delta_softplus)), batch, dim, N, L, G,
                                  _p(du), _p(dd), _p(dA), _p(dB), _p(dC), _p(dD), _p(db), _p(dz))
    if rc:
        raise RuntimeError(f"ms_oracle_scan_bwd failed rc={rc}")
    return dict(du=du, ddelta=dd, dA=dA, dB=dB[:, 0] if sq_b else dB, dC=dC[:, 0] if sq_c else dC,
                dD=dD, ddelta_bias=db, dz=dz)


def cross_scan(x):
    """(B,D,H,W) -> (B,4,D,H*W): MedMamba.py:393-395."""
    x = _c(x)
    b, d, h, w = x.shape
    xs = np.empty((b, 4, d, h * w), np.float32)
    lib().ms_oracle_cross_scan(_p(x), b, d, h, w, _p(xs))
    return xs


def cross_merge(ys, H, W):
    """(B,4,D,L) -> (B,D,L) = y1+y2+y3+y4 of MedMamba.py:420-424,476."""
    ys = _c(ys)
    b, k, d, L = ys.shape
    assert k == 4 and L == H * W
    y = np.empty((b, d, L), np.float32)
    lib().ms_oracle_cross_merge(_p(ys), b, d, H, W, _p(y))
    return y


# ------------------------------------------------------------------------------------------------
# torch wrapper (CPU tensors only) with the reference operator signature
# ------------------------------------------------------------------------------------------------
def _make_torch_fn():
    import torch

    class OracleSelectiveScanFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                    return_last_state=False):
            assert not u.is_cuda, "the oracle is a CPU checker"
            npz = lambda t: None if t is None else t.detach().float().contiguous().numpy()
            out, last = scan_fwd(npz(u), npz(delta), npz(A), npz(B), npz(C), npz(D), npz(z), npz(delta_bias),
                                 delta_softplus)
            ctx.save_for_backward(u, delta, A, B, C, D, z, delta_bias)
            ctx.delta_softplus = delta_softplus
            out_t = torch.from_numpy(out).to(u.dtype)
            if return_last_state:
                return out_t, torch.from_numpy(last)
            return out_t

        @staticmethod
        def backward(ctx, dout, *args):
            u, delta, A, B, C, D, z, delta_bias = ctx.saved_tensors
            npz = lambda t: None if t is None else t.detach().float().contiguous().numpy()
            g = scan_bwd(npz(u), npz(delta), npz(A), npz(B), npz(C), npz(D), npz(z), npz(delta_bias), npz(dout),
                         ctx.delta_softplus)
            t = lambda a, like: None if a is None else torch.from_numpy(a).to(like.dtype)
            return (t(g["du"], u), t(g["ddelta"], delta), t(g["dA"], A), t(g["dB"], B), t(g["dC"], C),
                    t(g["dD"], D) if D is not None else None, t(g["dz"], z) if z is not None else None,
                    t(g["ddelta_bias"], delta_bias) if delta_bias is not None else None, None, None)

    return OracleSelectiveScanFn


_FN = None


def selective_scan_oracle(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                          return_last_state=False):
    """torch front-end, same signature as the reference `selective_scan_fn` (selective_scan_interface.py:83-89)."""
    global _FN
    if _FN is None:
        _FN = _make_torch_fn()
    return _FN.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)
