"""Selective-scan operator -- drop-in for the reference's
`mamba_ssm.ops.selective_scan_interface` (CrossMamba/FusionMamba/mamba_ssm/ops/selective_scan_interface.py):
same names (`SelectiveScanFn`, `selective_scan_fn`), same argument meaning, same return values, same
error class (RuntimeError for bad operands), backed by the hand-written gfx950 kernels of libmedscan.so
through the C ABI in include/medscan.h instead of the CUDA extension `selective_scan_cuda`.

Differences that are deliberate and documented in DESIGN.md:
  * kernels do fp32 I/O; fp16/bf16 operands are up-cast on entry and the result is cast back (the reference
    kernels also compute in fp32 for every I/O dtype, selective_scan_fwd_kernel.cuh:147-160);
  * `z` gating is applied by two elementwise torch ops around the kernel (SS2D passes z=None, MedMamba.py:413);
  * complex A is rejected (never used by this repo's models);
  * the saved-state tensor `x` has layout (batch, n_chunks, dstate, dim) with 32-position chunks
    (private to the extension in the reference as well, selective_scan.cpp:307-313).
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import MsScanBwdParams, MsScanParams


class KernelTimer:
    """Optional HIP-event timing of the scan launches (bench.py's roofline leg).  Events are recorded on the
    stream the kernel is launched on, immediately before and after the launch; nothing synchronises until
    `summary()` is called after the timed region."""

    def __init__(self):
        self.enabled = False
        self.records = []          # (kind, algorithmic_bytes, state_elements, start_event, end_event)

    def launch(self, kind, nbytes, device, fn, state_elems=0):
        """state_elems = batch * dim * seqlen * dstate of the launch (the unit of the kernels' VALU work)."""
        if not self.enabled:
            return fn()
        s = torch.cuda.current_stream(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        rc = fn()
        e1.record(s)
        self.records.append((kind, nbytes, state_elems, e0, e1))
        return rc

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, nbytes, elems, e0, e1 in self.records:
            d = out.setdefault(kind, {"launches": 0, "ms": 0.0, "bytes": 0, "state_elems": 0})
            d["launches"] += 1; d["ms"] += e0.elapsed_time(e1); d["bytes"] += nbytes; d["state_elems"] += elems
        self.records = []
        return out


TIMER = KernelTimer()


def algorithmic_bytes(batch, dim, L, N, G, backward):
    """fp32, every operand of the operator touched once.
    fwd: u, delta in, out (3*B*dim*L) + B, C (2*B*G*N*L) + A, D, bias (dim*(N+2))
    bwd: u, delta, dout in, du, ddelta out (5*B*dim*L) + B, C in, dB, dC out (4*B*G*N*L)"""
    if backward:
        return 4 * (5 * batch * dim * L + 4 * batch * G * N * L)
    return 4 * (3 * batch * dim * L + 2 * batch * G * N * L + dim * (N + 2))


def _act_strides(t, dpg):
    """(batch, group, d, l) element strides of a (batch, dim, L) activation in the reference layout: group g
    starts dpg channels after group g-1."""
    sb, sd, sl = t.stride()
    return sb, dpg * sd, sd, sl


_SLICE = 16     # states per launch: the backward kernels keep <= 16 states of a channel in registers


def _state_slices(N):
    """The state axis in slices of <= 16 states.  The recurrence is independent per state and y / du / ddelta are sums
    over the states, so dstate up to the reference's 256 (selective_scan.cpp:262) is covered by ceil(N / 16) launches whose
    results add up (MS_SCAN_ACCUMULATE from the second slice on) -- what `mamba_chunk_scan_combined` in cnn_mamba.py does
    for the SSD blocks."""
    return [(s0, min(N, s0 + _SLICE)) for s0 in range(0, N, _SLICE)]


def _fill_fwd(P, u, delta, A, B, C, D, delta_bias, out, x, delta_softplus, accumulate=False):
    batch, dim, L = u.shape
    P.batch, P.dim, P.seqlen, P.dstate, P.n_groups = batch, dim, L, A.shape[1], B.shape[1]
    P.delta_softplus = int(bool(delta_softplus)) | (4 if accumulate else 0)      # MS_SCAN_SOFTPLUS | MS_SCAN_ACCUMULATE
    P.map_h = P.map_w = 0
    dpg = dim // B.shape[1]
    P.u_batch_stride, P.u_group_stride, P.u_d_stride, P.u_l_stride = _act_strides(u, dpg)
    P.delta_batch_stride, P.delta_group_stride, P.delta_d_stride, P.delta_l_stride = _act_strides(delta, dpg)
    if out is not None:
        P.out_batch_stride, P.out_group_stride, P.out_d_stride, P.out_l_stride = _act_strides(out, dpg)
    P.A_d_stride, P.A_dstate_stride = A.stride()
    P.B_batch_stride, P.B_group_stride, P.B_dstate_stride, P.B_l_stride = B.stride()
    P.C_batch_stride, P.C_group_stride, P.C_dstate_stride, P.C_l_stride = C.stride()
    P.u, P.delta, P.A, P.B, P.C = u.data_ptr(), delta.data_ptr(), A.data_ptr(), B.data_ptr(), C.data_ptr()
    P.D = D.data_ptr() if D is not None else None
    P.delta_bias = delta_bias.data_ptr() if delta_bias is not None else None
    P.out = out.data_ptr() if out is not None else None
    P.x = x.data_ptr() if x is not None else None


def _check_operands(u, delta, A, B, C, D, z, delta_bias):
    """Operand checks of selective_scan_fwd (selective_scan.cpp:233-303), raised as RuntimeError."""
    _lib.require_cuda(u, delta, A, B, C, D, z, delta_bias)
    if u.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise RuntimeError(f"selective_scan: unsupported input dtype {u.dtype}")
    if A.is_complex():
        raise RuntimeError("selective_scan: complex A is not supported by the MI355X kernels")
    if A.dtype != torch.float32:
        raise RuntimeError("selective_scan: A must be float32")
    if delta.dtype != u.dtype:
        raise RuntimeError("selective_scan: delta must have the dtype of u")
    if u.dim() != 3 or delta.shape != u.shape:
        raise RuntimeError(f"selective_scan: u/delta must be (batch, dim, seqlen); got {tuple(u.shape)} / {tuple(delta.shape)}")
    batch, dim, L = u.shape
    if A.dim() != 2 or A.shape[0] != dim:
        raise RuntimeError(f"selective_scan: A must be (dim, dstate); got {tuple(A.shape)}")
    N = A.shape[1]
    if N > 256:
        raise RuntimeError("selective_scan only supports state dimension <= 256")
    for name, t in (("B", B), ("C", C)):
        if t.dim() == 2:
            if tuple(t.shape) != (dim, N):
                raise RuntimeError(f"selective_scan: constant {name} must be (dim, dstate)")
        elif t.dim() == 3:
            if tuple(t.shape) != (batch, N, L):
                raise RuntimeError(f"selective_scan: {name} must be (batch, dstate, seqlen); got {tuple(t.shape)}")
        elif t.dim() == 4:
            if t.shape[0] != batch or t.shape[2] != N or t.shape[3] != L or dim % t.shape[1] != 0:
                raise RuntimeError(f"selective_scan: {name} must be (batch, groups, dstate, seqlen); got {tuple(t.shape)}")
        else:
            raise RuntimeError(f"selective_scan: bad rank for {name}")
    if B.dim() == 4 and C.dim() == 4 and B.shape[1] != C.shape[1]:
        raise RuntimeError("selective_scan: B and C must have the same number of groups")
    for name, t in (("D", D), ("delta_bias", delta_bias)):
        if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != (dim,)):
            raise RuntimeError(f"selective_scan: {name} must be float32 of shape (dim,)")
    if z is not None and (z.shape != u.shape or z.dtype != u.dtype):
        raise RuntimeError("selective_scan: z must match u")


def _as_groups(t, batch, dim, N, L):
    """B/C in any accepted form -> a 4-D (batch, G, N, L) fp32 tensor (possibly a stride-0 view)."""
    t = t.float()
    if t.dim() == 2:                       # constant (dim, N): one group per channel, broadcast over b and l
        return t.view(1, dim, N, 1).expand(batch, dim, N, L)
    if t.dim() == 3:
        t = t.unsqueeze(1)
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        t = t.contiguous()
    return t


class SelectiveScanFn(torch.autograd.Function):
    """Same contract as the reference class (selective_scan_interface.py:20-80)."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False):
        _check_operands(u, delta, A, B, C, D, z, delta_bias)
        lib = _lib.lib()
        in_dtype = u.dtype
        uf, df = u.float(), delta.float()
        if uf.stride(-1) != 1 and uf.stride(1) != 1:
            uf = uf.contiguous()
        if df.stride(-1) != 1 and df.stride(1) != 1:
            df = df.contiguous()
        batch, dim, L = uf.shape
        N = A.shape[1]
        ctx.b_shape, ctx.c_shape = tuple(B.shape), tuple(C.shape)
        ctx.b_dtype, ctx.c_dtype = B.dtype, C.dtype
        Bg, Cg = _as_groups(B, batch, dim, N, L), _as_groups(C, batch, dim, N, L)
        if Bg.shape[1] != Cg.shape[1]:     # one constant, one variable: bring both to per-channel groups
            G = max(Bg.shape[1], Cg.shape[1])
            Bg = Bg if Bg.shape[1] == G else Bg.repeat_interleave(G // Bg.shape[1], dim=1)
            Cg = Cg if Cg.shape[1] == G else Cg.repeat_interleave(G // Cg.shape[1], dim=1)
        # an A whose states are a stride-0 broadcast of one value per channel (SSD form) is passed through as is: the
        # kernels then evaluate one decay per position instead of one per state
        Af = A if (A.dim() == 2 and A.shape[1] > 1 and A.stride(1) == 0) else A.contiguous()
        Dc = D.contiguous() if D is not None else None
        bc = delta_bias.contiguous() if delta_bias is not None else None
        out = torch.empty_like(df)
        n_chunks = lib.ms_scan_n_chunks(L)
        slices = _state_slices(N)
        # saved states per slice: (n_slices, batch, n_chunks, <=16, dim); one slice = the reference-sized problem
        x = torch.empty((len(slices), batch, n_chunks, min(N, _SLICE), dim), device=u.device, dtype=torch.float32)
        if batch > 0 and L > 0:
            for i, (s0, s1) in enumerate(slices):
                P = MsScanParams()
                xi = x[i] if s1 - s0 == x.shape[3] else x[i].view(-1)[:batch * n_chunks * (s1 - s0) * dim].view(batch, n_chunks, s1 - s0, dim)
                _fill_fwd(P, uf, df, Af[:, s0:s1], Bg[:, :, s0:s1], Cg[:, :, s0:s1], Dc if i == 0 else None, bc, out, xi,
                          delta_softplus, accumulate=i > 0)
                with _lib.on_device(u.device):
                    rc = TIMER.launch("scan_fwd", algorithmic_bytes(batch, dim, L, s1 - s0, Bg.shape[1], False), u.device,
                                      lambda: lib.ms_selective_scan_fwd(ctypes.byref(P), _lib.current_stream_ptr(u.device)))
                    _lib.check(rc, "ms_selective_scan_fwd")
        ctx.delta_softplus = bool(delta_softplus)
        ctx.has_z = z is not None
        ctx.in_dtype = in_dtype
        ctx.has_D, ctx.has_bias = D is not None, delta_bias is not None
        if n_chunks > 0:
            last_state = torch.cat([(x[i] if s1 - s0 == x.shape[3] else
                                     x[i].view(-1)[:batch * n_chunks * (s1 - s0) * dim].view(batch, n_chunks, s1 - s0, dim))[:, -1]
                                    for i, (s0, s1) in enumerate(slices)], dim=1).transpose(1, 2)
        else:
            last_state = uf.new_zeros((batch, dim, N))
        if z is None:
            ctx.save_for_backward(uf, df, Af, Bg, Cg, Dc, bc, x)
            res = out.to(in_dtype)
        else:
            ctx.save_for_backward(uf, df, Af, Bg, Cg, Dc, bc, x, z, out)
            res = (out * F.silu(z.float())).to(in_dtype)
        return res if not return_last_state else (res, last_state)

    @staticmethod
    def backward(ctx, dout, *args):
        lib = _lib.lib()
        if ctx.has_z:
            uf, df, Af, Bg, Cg, Dc, bc, x, z, out = ctx.saved_tensors
            zf = z.float()
            sig = torch.sigmoid(zf)
            g = dout.float()
            dz = (g * out * (sig * (1 + zf * (1 - sig)))).to(z.dtype)
            g = g * zf * sig
        else:
            uf, df, Af, Bg, Cg, Dc, bc, x = ctx.saved_tensors
            g, dz = dout.float(), None
        if g.stride(-1) != 1 and g.stride(1) != 1:
            g = g.contiguous()
        batch, dim, L = uf.shape
        N, G = Af.shape[1], Bg.shape[1]
        du, ddelta = torch.empty_like(uf), torch.empty_like(df)
        dA = torch.zeros(Af.shape, device=Af.device, dtype=Af.dtype)       # dense even when Af is a broadcast view
        dB = torch.zeros((batch, G, N, L), device=uf.device, dtype=torch.float32)
        dC = torch.zeros_like(dB)
        dD = torch.zeros_like(Dc) if Dc is not None else None
        dbias = torch.zeros_like(bc) if bc is not None else None
        n_chunks = x.shape[2]
        for i, (s0, s1) in enumerate(_state_slices(N) if (batch > 0 and L > 0) else []):
            Q = MsScanBwdParams()
            xi = x[i] if s1 - s0 == x.shape[3] else x[i].view(-1)[:batch * n_chunks * (s1 - s0) * dim].view(batch, n_chunks, s1 - s0, dim)
            _fill_fwd(Q.f, uf, df, Af[:, s0:s1], Bg[:, :, s0:s1], Cg[:, :, s0:s1], Dc if i == 0 else None, bc, None, xi,
                      ctx.delta_softplus, accumulate=i > 0)
            dpg = dim // G
            Q.dout_batch_stride, Q.dout_group_stride, Q.dout_d_stride, Q.dout_l_stride = _act_strides(g, dpg)
            Q.du_batch_stride, Q.du_group_stride, Q.du_d_stride, Q.du_l_stride = _act_strides(du, dpg)
            Q.ddelta_batch_stride, Q.ddelta_group_stride, Q.ddelta_d_stride, Q.ddelta_l_stride = _act_strides(ddelta, dpg)
            Q.dB_batch_stride, Q.dB_group_stride, Q.dB_dstate_stride, Q.dB_l_stride = dB.stride()
            Q.dC_batch_stride, Q.dC_group_stride, Q.dC_dstate_stride, Q.dC_l_stride = dC.stride()
            Q.dout, Q.du, Q.ddelta = g.data_ptr(), du.data_ptr(), ddelta.data_ptr()
            # the kernels write dA as a dense (dim, states of this launch) block: one scratch block per slice
            dAi = dA if len(_state_slices(N)) == 1 else torch.zeros((dim, s1 - s0), device=Af.device, dtype=torch.float32)
            Q.dA, Q.dB, Q.dC = dAi.data_ptr(), dB[:, :, s0:s1].data_ptr(), dC[:, :, s0:s1].data_ptr()
            Q.dD = dD.data_ptr() if (dD is not None and i == 0) else None
            Q.ddelta_bias = dbias.data_ptr() if dbias is not None else None
            with _lib.on_device(uf.device):
                rc = TIMER.launch("scan_bwd", algorithmic_bytes(batch, dim, L, s1 - s0, G, True), uf.device,
                                  lambda: lib.ms_selective_scan_bwd(ctypes.byref(Q), _lib.current_stream_ptr(uf.device)))
                _lib.check(rc, "ms_selective_scan_bwd")
            if dAi is not dA:
                dA[:, s0:s1] = dAi

        def back_to(grad, shape, dtype):   # undo _as_groups
            if len(shape) == 2:
                grad = grad.sum(dim=(0, 3)) if grad.shape[1] == shape[0] else grad
            elif len(shape) == 3:
                # (batch, N, L) operand: one group -- or `dim` groups when the OTHER operand was a per-channel constant
                # (both were expanded to per-channel groups): sum the expanded axis back
                grad = grad.squeeze(1) if grad.shape[1] == 1 else grad.sum(dim=1)
            elif grad.shape[1] != shape[1]:
                grad = grad.view(shape[0], shape[1], -1, shape[2], shape[3]).sum(2)
            return grad.to(dtype)

        return (du.to(ctx.in_dtype), ddelta.to(ctx.in_dtype), dA, back_to(dB, ctx.b_shape, ctx.b_dtype),
                back_to(dC, ctx.c_shape, ctx.c_dtype), dD, dz, dbias, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """if return_last_state is True, returns (out, last_state); last_state is (batch, dim, dstate) and
    carries no gradient -- same contract as selective_scan_interface.py:83-89."""
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)
