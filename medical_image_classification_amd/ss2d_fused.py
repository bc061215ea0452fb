"""Channel-last fused core of SS2D (what sits between in_proj and out_norm in MedMamba.py:466-483).

The reference permutes to NCHW, materialises the four scan orders (stack/transpose/flip/cat), runs the projections on
those copies, scans, and un-permutes/merges with more copies (MedMamba.py:393-424,472-477).  x_proj and dt_proj are
pointwise per pixel, so here every tensor stays in PIXEL order, channel-last:

    xc    (B,H,W,D)   = SiLU(dwconv3x3(x) + b)                      ms_dwconv3x3_silu_nhwc_*  (reads xz in place)
    P     (B,L,4,C)   = xc @ x_proj_weight.view(4C, D)^T            C = R + 2N: [dts | Bs | Cs] of every direction
    delta (4,B,L,D)   = P[..., k, :R] @ dt_projs_weight[k]^T        (bias + softplus inside the scan kernel)
    y     (B,L,D)     = sum_k scan_k(...)                           ms_selective_scan_fwd in SS2D mode: direction k visits
                                                                    the pixels in the order of MedMamba.py:393-395, all
                                                                    four read the same xc (group stride 0) and write
                                                                    their result back in pixel order
so cross-scan and cross-merge cost no memory traffic at all, and nothing is ever transposed.
"""
import ctypes
import os

import torch

from . import _lib, arena, shadow
from ._lib import MsScanBwdParams, MsScanParams
from .gemm_ops import gemm, gemm_f32, linear_bwd_fused, linear_bwd_fused_ok, weight_grad, weight_grad_f32
from .selective_scan_interface import TIMER, algorithmic_bytes
from .ss2d_ops import _F32_GEMM, _MFMA_GEMM, _MFMA_MIN_ROWS


class _DWConvSiLUNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        """x: (B,H,W,C) view with unit channel stride and a uniform pixel stride (e.g. one half of xz); fp32 or bf16."""
        _lib.require_cuda(x, weight, bias)
        B, H, W, C = x.shape
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        ps = x.stride(2)
        if x.stride(3) != 1 or x.stride(1) != W * ps or x.stride(0) != H * W * ps:
            x = x.contiguous(); ps = C
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous() if bias is not None else None
        y = torch.empty((B, H, W, C), device=x.device, dtype=torch.float32)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ms_dwconv3x3_silu_nhwc_fwd(
                x.data_ptr(), int(x.dtype == torch.bfloat16), w.data_ptr(), b.data_ptr() if b is not None else None,
                y.data_ptr(), B, C, H, W, ps, _lib.current_stream_ptr(x.device)), "ms_dwconv3x3_silu_nhwc_fwd")
        ctx.save_for_backward(x, w, b)
        ctx.ps, ctx.wdtype, ctx.bdtype = ps, weight.dtype, (bias.dtype if bias is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        B, H, W, C = x.shape
        dy = dy.contiguous().float()
        dx = torch.empty((B, H, W, C), device=x.device, dtype=torch.float32)
        scratch = _dwconv_bwd_scratch(B, C, H, W, x.device)
        dw = arena.zeros_like(w)
        db = arena.zeros_like(b) if b is not None else None
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ms_dwconv3x3_silu_nhwc_bwd(
                x.data_ptr(), int(x.dtype == torch.bfloat16), w.data_ptr(), b.data_ptr() if b is not None else None,
                dy.data_ptr(), 1, 0, None, dx.data_ptr(), 0, C, scratch.data_ptr(), dw.data_ptr(),
                db.data_ptr() if db is not None else None, B, C, H, W, ctx.ps, _lib.current_stream_ptr(x.device)),
                "ms_dwconv3x3_silu_nhwc_bwd")
        return dx.to(x.dtype), dw.to(ctx.wdtype), (db.to(ctx.bdtype) if db is not None else None)


def _dwconv_bwd_scratch(B, C, H, W, device):
    """Workspace of ms_dwconv3x3_silu_nhwc_bwd: the pre-activation gradient + the per-workgroup dw/dbias partial sums."""
    return torch.empty(_lib.lib().ms_dwconv3x3_silu_nhwc_bwd_scratch_floats(B, C, H, W), device=device, dtype=torch.float32)


def dwconv3x3_silu_nhwc(x, weight, bias):
    """SiLU(depthwise conv3x3(x) + bias) on a channel-last (B,H,W,C) tensor -> (B,H,W,C) fp32."""
    return _DWConvSiLUNHWC.apply(x, weight, bias)


_SEG_TARGET_WAVES = int(os.environ.get("MEDSCAN_SCAN_SEG_WAVES", "2048"))      # 0: never segment


def _scan_segments(waves, n_chunks):
    """Segments of the sequence for an inference forward scan: enough to put ~_SEG_TARGET_WAVES waves on the chip, at least 4 chunks
    (128 positions) per segment; 1 = unsegmented."""
    if _SEG_TARGET_WAVES <= 0 or waves >= _SEG_TARGET_WAVES // 2 or n_chunks < 8:
        return 1
    return max(1, min(n_chunks // 4, -(-_SEG_TARGET_WAVES // waves)))


def _ss2d_params(P, xc, proj, delta, A, Ds, dt_bias, out, x_state, H, W, N, R, a_is_log=False):
    B, L, D = xc.shape[0], H * W, xc.shape[-1]
    C = R + 2 * N
    P.batch, P.dim, P.seqlen, P.dstate, P.n_groups = B, 4 * D, L, N, 4
    P.delta_softplus, P.map_h, P.map_w = 1 | (2 if a_is_log else 0), H, W      # MS_SCAN_SOFTPLUS | MS_SCAN_A_IS_LOG
    P.u_batch_stride, P.u_group_stride, P.u_d_stride, P.u_l_stride = L * D, 0, 1, D
    P.delta_batch_stride, P.delta_group_stride, P.delta_d_stride, P.delta_l_stride = L * D, B * L * D, 1, D
    P.out_batch_stride, P.out_group_stride, P.out_d_stride, P.out_l_stride = L * D, B * L * D, 1, D
    P.A_d_stride, P.A_dstate_stride = N, 1
    P.B_batch_stride, P.B_group_stride, P.B_dstate_stride, P.B_l_stride = L * 4 * C, C, 1, 4 * C
    P.C_batch_stride, P.C_group_stride, P.C_dstate_stride, P.C_l_stride = L * 4 * C, C, 1, 4 * C
    P.u, P.delta, P.A = xc.data_ptr(), (delta.data_ptr() if delta is not None else None), A.data_ptr()
    P.B, P.C = proj.data_ptr() + 4 * R, proj.data_ptr() + 4 * (R + N)
    P.D, P.delta_bias = Ds.data_ptr(), dt_bias.data_ptr()
    P.out = out.data_ptr() if out is not None else None
    P.x = x_state.data_ptr() if x_state is not None else None


# dt_rank up to which ms_dtproj_* are used (above it: batched GEMMs).  R <= 4: the register-tiled kernels (stage 0 of MedMamba-T,
# 284 vs 438 us fwd+bwd for the GEMMs); R = 5..32: the scalar-operand kernels (csrc/dtproj.hip), measured in tools/bench_dtproj.py.
_DT_KERNEL_MAX_RANK = int(os.environ.get("MEDSCAN_DT_KERNEL_MAX_RANK", "32"))


def _split_k(M):
    for cand in (64, 32, 16, 8, 4, 2):
        if M % cand == 0 and M // cand >= 1024:
            return cand
    return 1


_DT_ACT = os.environ.get("MEDSCAN_DT_ACTIVATED", "1") == "1"
# training forward: delta' formed inside the scan kernel and stored for the backward (MS_SCAN_DELTA_OUT; no ms_dtproj_fwd_act launch).
# Measured on MI355X (MedMamba-T bs 64): 17.037 vs 17.039 ms per step -- the R FMAs + softplus per element that move into the
# issue-bound forward scan cost what the 8 projection launches cost -- so it stays opt-in (parity: test_modules_gpu with =1)
_DT_FUSED_TRAIN = os.environ.get("MEDSCAN_DT_FUSED_TRAIN", "0") == "1"


def _dtproj_fwd(proj, wdt, B, L, D, R, C, act_bias=None):
    """delta (4,B,L,D) = dts @ Wdt^T (MedMamba.py:400), fp32, with dts = the first R columns of the projection rows:
    ms_dtproj_fwd (reads them in place), one batched GEMM above _DT_KERNEL_MAX_RANK.  Runs inside the scan's autograd
    node, so neither the dts slice nor delta ever becomes an autograd tensor.
    act_bias (4*D,): ms_dtproj_fwd_act -- the kernel (bandwidth-bound) also applies delta' = softplus(delta + bias), which the
    issue-bound scan kernels then skip (MS_SCAN_DELTA_ACTIVATED); only the kernel path can, so callers check `_dt_act_ok(R)`."""
    M = B * L
    if R <= _DT_KERNEL_MAX_RANK:
        delta = torch.empty((4, B, L, D), device=proj.device, dtype=torch.float32)
        if act_bias is not None:
            _lib.check(_lib.lib().ms_dtproj_fwd_act(proj.data_ptr(), wdt.data_ptr(), act_bias.data_ptr(), delta.data_ptr(), M, D, R, C,
                                                    _lib.current_stream_ptr(proj.device)), "ms_dtproj_fwd_act")
        else:
            _lib.check(_lib.lib().ms_dtproj_fwd(proj.data_ptr(), wdt.data_ptr(), delta.data_ptr(), M, D, R, C,
                                                _lib.current_stream_ptr(proj.device)), "ms_dtproj_fwd")
        return delta
    assert act_bias is None
    dts = proj.view(M, 4, C)[:, :, :R].permute(1, 0, 2)          # (4, M, R) view: row stride 4C, unit inner stride -- a valid GEMM operand
    return torch.bmm(dts, wdt.transpose(1, 2)).view(4, B, L, D)


def _dt_act_ok(R):
    return _DT_ACT and R <= _DT_KERNEL_MAX_RANK


def _dtproj_bwd(ddelta, proj, wdt, dproj, B, L, D, R, C):
    """ddts into the first R columns of dproj (in place: the scan backward owns the B|C columns), returns dWdt (4,D,R).
    The weight gradient reduces over M = B*L with a (D x R) output: split into slices, one batched GEMM, fp32 sum."""
    M = B * L
    if R <= _DT_KERNEL_MAX_RANK:
        dwdt = arena.zeros_like(wdt)
        ns = _lib.lib().ms_dtproj_bwd_scratch_floats(M, D, R)
        scratch = torch.empty(ns, device=proj.device, dtype=torch.float32) if ns > 0 else None
        _lib.check(_lib.lib().ms_dtproj_bwd(ddelta.data_ptr(), proj.data_ptr(), wdt.data_ptr(), dproj.data_ptr(), dwdt.data_ptr(),
                                            scratch.data_ptr() if scratch is not None else None, ns,
                                            M, D, R, C, _lib.current_stream_ptr(proj.device)), "ms_dtproj_bwd")
        return dwdt
    dd = ddelta.view(4, M, D)
    dproj.view(M, 4, C)[:, :, :R].copy_(torch.bmm(dd, wdt).permute(1, 0, 2))
    dts = proj.view(M, 4, C)[:, :, :R].permute(1, 0, 2).contiguous()
    S = _split_k(M)
    part = torch.bmm(dd.view(4 * S, M // S, D).transpose(1, 2), dts.view(4 * S, M // S, R))   # (4S, D, R)
    return part.view(4, S, D, R).sum(dim=1)


class _SS2DScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xc, proj, delta, wdt, A, Ds, dt_bias, H, W, N, R, lattice=False):
        """lattice=True: the four groups scan the four stride-2 sub-lattices of the (even-sized) map instead of the four
        full-resolution directions (MS_SCAN_LATTICE: FusionMamba's EfficientScan / EfficientMerge, cross.py:139-190,34-88, as an
        addressing mode): sequences of H*W/4 pixels, every pixel written by exactly one group -> y IS the merged result.
        xc (B,H,W,D), proj (B,L,4,R+2N), A (4D,N), Ds (4D), dt_bias (4D): fp32 contiguous; either delta (4,B,L,D) or
        wdt (4,D,R) (then delta = dts @ wdt^T is computed here, ms_dtproj_fwd).
        Returns y (B,L,D) = ((y0 + y2) + y1) + y3 with every y_k in pixel order (add order of MedMamba.py:476)."""
        _lib.require_cuda(xc, proj, A, Ds, dt_bias)
        lib = _lib.lib()
        B, D, L = xc.shape[0], xc.shape[-1], H * W
        xc, proj = xc.contiguous(), proj.contiguous()
        A, Ds, dt_bias = A.contiguous(), Ds.contiguous(), dt_bias.contiguous()
        ctx.has_wdt = wdt is not None
        with _lib.on_device(xc.device):
            if ctx.has_wdt:
                wdt = wdt.detach().float().contiguous()
                delta = _dtproj_fwd(proj, wdt, B, L, D, R, R + 2 * N)
            else:
                delta = delta.contiguous()
        if lattice and (H % 2 or W % 2):
            raise RuntimeError("ss2d scan, lattice mode: the map must have even sizes (zero-pad it, cross.py:148-156)")
        Lq = L // 4 if lattice else L                    # sequence length of one group
        y4 = torch.empty((B, L, D) if lattice else (4, B, L, D), device=xc.device, dtype=torch.float32)
        x_state = torch.empty((B, lib.ms_scan_n_chunks(Lq), N, 4 * D), device=xc.device, dtype=torch.float32)
        P = MsScanParams()
        _ss2d_params(P, xc, proj, delta, A, Ds, dt_bias, y4, x_state, H, W, N, R)
        if lattice:
            P.seqlen, P.out_group_stride = Lq, 0
            P.delta_softplus |= 256                       # MS_SCAN_LATTICE
        with _lib.on_device(xc.device):
            rc = TIMER.launch("scan_fwd", algorithmic_bytes(B, 4 * D, Lq, N, 4, False), xc.device,
                              lambda: lib.ms_selective_scan_fwd(ctypes.byref(P), _lib.current_stream_ptr(xc.device)))
            _lib.check(rc, "ms_selective_scan_fwd[ss2d]")
        ctx.save_for_backward(xc, proj, delta, A, Ds, dt_bias, x_state, wdt if ctx.has_wdt else None)
        ctx.geom = (H, W, N, R, bool(lattice))
        return y4 if lattice else (y4[0] + y4[2]) + y4[1] + y4[3]

    @staticmethod
    def backward(ctx, dy):
        xc, proj, delta, A, Ds, dt_bias, x_state, wdt = ctx.saved_tensors
        H, W, N, R, lattice = ctx.geom
        lib = _lib.lib()
        B, D, L = xc.shape[0], xc.shape[-1], H * W
        C = R + 2 * N
        dy = dy.contiguous().float()                              # (B,L,D): dout of all four directions (group stride 0)
        # lattice: every pixel belongs to one group -> ONE du tensor (group stride 0); ddelta keeps its four slabs (the
        # projection backward sums over all of them), of which a group writes only its own pixels: zero-filled first
        du4 = torch.empty((B, L, D) if lattice else (4, B, L, D), device=xc.device, dtype=torch.float32)
        ddelta = torch.zeros((4, B, L, D), device=xc.device, dtype=torch.float32) if lattice else torch.empty_like(du4)
        dproj = arena.zeros_like(proj)
        dA, dD, dbias = arena.zeros_like(A), arena.zeros_like(Ds), arena.zeros_like(dt_bias)
        Q = MsScanBwdParams()
        _ss2d_params(Q.f, xc, proj, delta, A, Ds, dt_bias, None, x_state, H, W, N, R)
        if lattice:
            Q.f.seqlen = L // 4
            Q.f.delta_softplus |= 256                     # MS_SCAN_LATTICE
        Q.dout_batch_stride, Q.dout_group_stride, Q.dout_d_stride, Q.dout_l_stride = L * D, 0, 1, D
        Q.du_batch_stride, Q.du_group_stride, Q.du_d_stride, Q.du_l_stride = L * D, (0 if lattice else B * L * D), 1, D
        Q.ddelta_batch_stride, Q.ddelta_group_stride, Q.ddelta_d_stride, Q.ddelta_l_stride = L * D, B * L * D, 1, D
        Q.dB_batch_stride, Q.dB_group_stride, Q.dB_dstate_stride, Q.dB_l_stride = L * 4 * C, C, 1, 4 * C
        Q.dC_batch_stride, Q.dC_group_stride, Q.dC_dstate_stride, Q.dC_l_stride = L * 4 * C, C, 1, 4 * C
        Q.dout, Q.du, Q.ddelta = dy.data_ptr(), du4.data_ptr(), ddelta.data_ptr()
        Q.dA, Q.dD, Q.ddelta_bias = dA.data_ptr(), dD.data_ptr(), dbias.data_ptr()
        Q.dB, Q.dC = dproj.data_ptr() + 4 * R, dproj.data_ptr() + 4 * (R + N)
        with _lib.on_device(xc.device):
            rc = TIMER.launch("scan_bwd", algorithmic_bytes(B, 4 * D, L, N, 4, True), xc.device,
                              lambda: lib.ms_selective_scan_bwd(ctypes.byref(Q), _lib.current_stream_ptr(xc.device)))
            _lib.check(rc, "ms_selective_scan_bwd[ss2d]")
            dwdt = _dtproj_bwd(ddelta, proj, wdt, dproj, B, L, D, R, C) if ctx.has_wdt else None
        dxc = du4.view_as(xc) if lattice else du4.sum(dim=0).view_as(xc)
        return dxc, dproj, (None if ctx.has_wdt else ddelta), dwdt, dA, dD, dbias, None, None, None, None, None


def _pixel_view(t, D):
    """(B,H,W,D) tensor (possibly a channel slice of a wider tensor) -> (tensor, pixel stride) usable by the kernels."""
    B, H, W, _ = t.shape
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    ps = t.stride(2)
    if t.stride(3) != 1 or t.stride(1) != W * ps or t.stride(0) != H * W * ps or ps < D:
        t = t.contiguous(); ps = D
    return t, ps


class _SS2DScanNormGate(torch.autograd.Function):
    """scan (4 directions, pixel order) -> merge sum + LayerNorm + SiLU(z) gate, as ONE autograd node: the four scan
    outputs never pass through autograd, and the LayerNorm backward hands one (B,L,D) gradient to all four directions."""

    @staticmethod
    def forward(ctx, xc, proj, delta, wdt, A, Ds, dt_bias, z, gamma, beta, eps, H, W, N, R, out_bf16):
        """`A` is A_logs (4D,N): the kernels apply A = -exp(A_logs) themselves and return d/dA_logs."""
        _lib.require_cuda(xc, proj, A, Ds, dt_bias, z, gamma, beta)
        lib = _lib.lib()
        B, D, L = xc.shape[0], xc.shape[-1], H * W
        xc, proj = xc.contiguous(), proj.contiguous()
        A, Ds, dt_bias = A.contiguous(), Ds.contiguous(), dt_bias.contiguous()
        ctx.has_wdt = wdt is not None
        with _lib.on_device(xc.device):
            if ctx.has_wdt:
                wdt = wdt.detach().float().contiguous()
                delta = _dtproj_fwd(proj, wdt, B, L, D, R, R + 2 * N)
            else:
                delta = delta.contiguous()
        gamma, beta = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        z, zps = _pixel_view(z, D)
        y4 = torch.empty((4, B, L, D), device=xc.device, dtype=torch.float32)
        x_state = torch.empty((B, lib.ms_scan_n_chunks(L), N, 4 * D), device=xc.device, dtype=torch.float32)
        out = torch.empty((B, H, W, D), device=xc.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        P = MsScanParams()
        _ss2d_params(P, xc, proj, delta, A, Ds, dt_bias, y4, x_state, H, W, N, R, a_is_log=True)
        stream = _lib.current_stream_ptr(xc.device)
        with _lib.on_device(xc.device):
            rc = TIMER.launch("scan_fwd", algorithmic_bytes(B, 4 * D, L, N, 4, False), xc.device,
                              lambda: lib.ms_selective_scan_fwd(ctypes.byref(P), stream))
            _lib.check(rc, "ms_selective_scan_fwd[ss2d]")
            _lib.check(lib.ms_ln_gate_fwd(y4.data_ptr(), B * L * D, z.data_ptr(), int(z.dtype == torch.bfloat16), zps,
                                          gamma.data_ptr(), beta.data_ptr(), float(eps), out.data_ptr(), int(out_bf16),
                                          B * L, D, stream), "ms_ln_gate_fwd")
        ctx.save_for_backward(xc, proj, delta, A, Ds, dt_bias, x_state, y4, z, gamma, beta, wdt if ctx.has_wdt else None)
        ctx.geom = (H, W, N, R, float(eps), zps)
        return out

    @staticmethod
    def backward(ctx, dout):
        xc, proj, delta, A, Ds, dt_bias, x_state, y4, z, gamma, beta, wdt = ctx.saved_tensors
        H, W, N, R, eps, zps = ctx.geom
        lib = _lib.lib()
        B, D, L = xc.shape[0], xc.shape[-1], H * W
        C = R + 2 * N
        if dout.dtype not in (torch.float32, torch.bfloat16):
            dout = dout.float()
        dout = dout.contiguous()
        dy = torch.empty((B, L, D), device=xc.device, dtype=torch.float32)
        dz = torch.empty((B, H, W, D), device=xc.device, dtype=z.dtype)
        du4 = torch.empty((4, B, L, D), device=xc.device, dtype=torch.float32)
        ddelta = torch.empty_like(du4)
        dproj = arena.zeros_like(proj)
        # the small accumulators share one zero-filled buffer (one fill launch instead of five)
        sizes = (A.numel(), Ds.numel(), dt_bias.numel(), gamma.numel(), beta.numel())
        zbuf = arena.zeros(sum(sizes), xc.device)
        dA, dD, dbias, dgamma, dbeta = (t.view(r.shape) for t, r in zip(zbuf.split(sizes), (A, Ds, dt_bias, gamma, beta)))
        Q = MsScanBwdParams()
        _ss2d_params(Q.f, xc, proj, delta, A, Ds, dt_bias, None, x_state, H, W, N, R, a_is_log=True)
        Q.dout_batch_stride, Q.dout_group_stride, Q.dout_d_stride, Q.dout_l_stride = L * D, 0, 1, D
        Q.du_batch_stride, Q.du_group_stride, Q.du_d_stride, Q.du_l_stride = L * D, B * L * D, 1, D
        Q.ddelta_batch_stride, Q.ddelta_group_stride, Q.ddelta_d_stride, Q.ddelta_l_stride = L * D, B * L * D, 1, D
        Q.dB_batch_stride, Q.dB_group_stride, Q.dB_dstate_stride, Q.dB_l_stride = L * 4 * C, C, 1, 4 * C
        Q.dC_batch_stride, Q.dC_group_stride, Q.dC_dstate_stride, Q.dC_l_stride = L * 4 * C, C, 1, 4 * C
        Q.dout, Q.du, Q.ddelta = dy.data_ptr(), du4.data_ptr(), ddelta.data_ptr()
        Q.dA, Q.dD, Q.ddelta_bias = dA.data_ptr(), dD.data_ptr(), dbias.data_ptr()
        Q.dB, Q.dC = dproj.data_ptr() + 4 * R, dproj.data_ptr() + 4 * (R + N)
        stream = _lib.current_stream_ptr(xc.device)
        with _lib.on_device(xc.device):
            _lib.check(lib.ms_ln_gate_bwd(y4.data_ptr(), B * L * D, z.data_ptr(), int(z.dtype == torch.bfloat16), zps,
                                          gamma.data_ptr(), beta.data_ptr(), eps, dout.data_ptr(),
                                          int(dout.dtype == torch.bfloat16), dy.data_ptr(), dz.data_ptr(), D,
                                          dgamma.data_ptr(), dbeta.data_ptr(), B * L, D, stream), "ms_ln_gate_bwd")
            rc = TIMER.launch("scan_bwd", algorithmic_bytes(B, 4 * D, L, N, 4, True), xc.device,
                              lambda: lib.ms_selective_scan_bwd(ctypes.byref(Q), stream))
            _lib.check(rc, "ms_selective_scan_bwd[ss2d]")
            dwdt = _dtproj_bwd(ddelta, proj, wdt, dproj, B, L, D, R, C) if ctx.has_wdt else None
        dxc = du4.sum(dim=0).view_as(xc)
        return (dxc, dproj, (None if ctx.has_wdt else ddelta), dwdt, dA, dD, dbias, dz, dgamma, dbeta,
                None, None, None, None, None, None)


class _SS2DInner(torch.autograd.Function):
    """Everything between in_proj and out_proj of SS2D (MedMamba.py:469-479) as ONE autograd node:
    depthwise conv + SiLU -> x_proj -> dt_proj -> 4-direction scan -> merge + out_norm + SiLU(z) gate.
    Backward hands gradients from kernel to kernel without autograd glue: the scan's four per-direction du slabs and the
    x_proj input gradient are summed inside the conv backward's load (no du4.sum, no gradient add), and the conv's dx and
    the gate's dz are written (in xz's dtype) straight into the two halves of ONE (B,H,W,2D) gradient buffer (no casts, no
    concat)."""

    @staticmethod
    def forward(ctx, xz, conv_w, conv_b, xproj_w, wdt, A_logs, Ds, dt_bias, gamma, beta, eps, N, R, out_bf16, mm_dtype, want_grad=True):
        _lib.require_cuda(xz, conv_w, xproj_w, wdt, A_logs, Ds, dt_bias, gamma, beta)
        lib = _lib.lib()
        B, H, W, D2 = xz.shape
        D, L, M, C = D2 // 2, H * W, B * H * W, R + 2 * N
        if xz.dtype not in (torch.float32, torch.bfloat16):
            xz = xz.float()
        xz = xz.contiguous()
        isz, xz_bf16 = xz.element_size(), int(xz.dtype == torch.bfloat16)
        f32 = lambda t: t.detach().float().contiguous()
        cw, cb = f32(conv_w), (f32(conv_b) if conv_b is not None else None)
        wdt, A, Dv, bias, gamma, beta = f32(wdt), f32(A_logs), f32(Ds), f32(dt_bias), f32(gamma), f32(beta)
        wx = xproj_w.detach().reshape(4 * C, D)
        mfma = mm_dtype == torch.bfloat16 and _MFMA_GEMM and D % 8 == 0 and M >= _MFMA_MIN_ROWS       # x_proj on ms_gemm_bf16 (fp32 operands read in place)
        if mfma and xproj_w.dtype == torch.float32 and xproj_w.is_contiguous():
            wx = shadow.bf16(xproj_w).view(4 * C, D)       # cached bf16 copy, refreshed with all the others by one launch per step
        else:
            wx = (wx.to(mm_dtype) if (mm_dtype is not None and not mfma) else wx.float()).contiguous()
        # fp32 runs (no autocast: the reference's precision): x_proj and its autograd on ms_gemm_f32, no library GEMM
        f32mm = mm_dtype is None and not mfma and _F32_GEMM and D % 4 == 0 and (4 * C) % 4 == 0
        dev = xz.device
        xc = torch.empty((B, H, W, D), device=dev, dtype=torch.float32)
        y4 = torch.empty((4, B, L, D), device=dev, dtype=torch.float32)
        x_state = torch.empty((B, lib.ms_scan_n_chunks(L), N, 4 * D), device=dev, dtype=torch.float32)
        out = torch.empty((B, H, W, D), device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        stream = _lib.current_stream_ptr(dev)
        with _lib.on_device(dev):
            _lib.check(lib.ms_dwconv3x3_silu_nhwc_fwd(xz.data_ptr(), xz_bf16, cw.data_ptr(), cb.data_ptr() if cb is not None else None,
                                                      xc.data_ptr(), B, D, H, W, D2, stream), "ms_dwconv3x3_silu_nhwc_fwd")
            if mfma:
                xm = None
                proj = gemm(xc.view(M, D), wx)                                                               # (M, 4C) fp32
            else:
                xm = xc.view(M, D).to(mm_dtype) if mm_dtype is not None else xc.view(M, D)
                if f32mm:
                    proj = gemm_f32(xm, wx)                                                                  # exact-fp32 MFMA kernel
                else:
                    proj = torch.mm(xm, wx.t(), out_dtype=torch.float32) if mm_dtype is not None else torch.mm(xm, wx.t())
            # inference (no gradient wanted): the Delta projection is formed inside the scan kernel (MS_SCAN_DT_FUSED) --
            # no dt_proj launch, no delta tensor, no saved states.  Training materialises delta: the backward kernel reads it.
            # `want_grad`: under torch.no_grad() `needs_input_grad` still mirrors the parameters' requires_grad, so the wrapper passes
            # whether a graph is being recorded at all (the validation loop of train.py:82-95 is eval() + no_grad())
            need_bwd = want_grad and any(ctx.needs_input_grad)
            fusable = N == 16 and R <= 32 and D % 4 == 0
            fuse_dt = (not need_bwd) and fusable
            # training: the forward scan forms delta' itself AND stores it for the backward launch (MS_SCAN_DELTA_OUT): no dt_proj
            # forward launch, no pre-activation tensor (MedMamba.py:400,403-405)
            fuse_dt_train = need_bwd and fusable and _DT_FUSED_TRAIN and _dt_act_ok(R)
            act = fuse_dt_train or ((not fuse_dt) and _dt_act_ok(R))   # delta' = softplus(delta + bias) reaches the backward activated
            if fuse_dt_train:
                delta = torch.empty((4, B, L, D), device=dev, dtype=torch.float32)
            else:
                delta = None if fuse_dt else _dtproj_fwd(proj, wdt, B, L, D, R, C, act_bias=bias if act else None)
            P = MsScanParams()
            _ss2d_params(P, xc, proj, delta, A, Dv, bias, y4, x_state, H, W, N, R, a_is_log=True)
            if act and not fuse_dt_train:
                P.delta_softplus |= 512             # MS_SCAN_DELTA_ACTIVATED
            if fuse_dt:
                P.delta_softplus |= 128             # MS_SCAN_DT_FUSED
                P.dt_x, P.dt_w, P.dt_rank, P.x = proj.data_ptr(), wdt.data_ptr(), R, None
            if not need_bwd:
                # inference at small batch: B * 4 * D / 8 waves walk all L positions one after the other and do not fill the chip (batch 1 at
                # stage 0 of MedMamba-T: 48 waves for 1 024 SIMDs) -- scan the sequence in segments (MsScanParams.segments, csrc/scan_ss2d.hip)
                segs = _scan_segments(B * 4 * ((D + 7) // 8), lib.ms_scan_n_chunks(L))
                if segs >= 2 and N == 16 and D % 4 == 0:
                    seg_ws = torch.empty(lib.ms_scan_seg_floats(B, 4 * D, segs), device=dev, dtype=torch.float32)
                    P.x, P.segments = seg_ws.data_ptr(), segs
            if fuse_dt_train:
                P.delta_softplus |= 128 | 1024      # MS_SCAN_DT_FUSED | MS_SCAN_DELTA_OUT
                P.dt_x, P.dt_w, P.dt_rank = proj.data_ptr(), wdt.data_ptr(), R
            rc = TIMER.launch("scan_fwd", algorithmic_bytes(B, 4 * D, L, N, 4, False), dev,
                              lambda: lib.ms_selective_scan_fwd(ctypes.byref(P), stream), B * 4 * D * L * N)
            _lib.check(rc, "ms_selective_scan_fwd[ss2d]")
            # training keeps the MERGED sum of the four direction slabs for the backward (ms_ln_gate_fwd_keep): the LayerNorm
            # backward then reads 4 B instead of 16 B per element and the slabs are released here
            ysum = torch.empty((B, L, D), device=dev, dtype=torch.float32) if need_bwd else None
            if ysum is None:
                _lib.check(lib.ms_ln_gate_fwd(y4.data_ptr(), B * L * D, xz.data_ptr() + D * isz, xz_bf16, D2, gamma.data_ptr(),
                                              beta.data_ptr(), float(eps), out.data_ptr(), int(out_bf16), M, D, stream),
                           "ms_ln_gate_fwd")
            else:
                _lib.check(lib.ms_ln_gate_fwd_keep(y4.data_ptr(), B * L * D, xz.data_ptr() + D * isz, xz_bf16, D2, gamma.data_ptr(),
                                                   beta.data_ptr(), float(eps), out.data_ptr(), int(out_bf16), ysum.data_ptr(),
                                                   M, D, stream), "ms_ln_gate_fwd_keep")
        ctx.save_for_backward(xz, xc, xm if (mm_dtype is not None and not mfma) else None, wx, proj, delta, x_state, ysum, cw, cb, wdt,
                              A, Dv, bias, gamma, beta)
        ctx.geom = (N, R, float(eps))
        ctx.mfma, ctx.act, ctx.f32mm = mfma, act, f32mm
        ctx.dtypes = (conv_w.dtype, conv_b.dtype if conv_b is not None else None, xproj_w.dtype, xproj_w.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        xz, xc, xm, wx, proj, delta, x_state, ysum, cw, cb, wdt, A, Dv, bias, gamma, beta = ctx.saved_tensors
        N, R, eps = ctx.geom
        lib = _lib.lib()
        B, H, W, D2 = xz.shape
        D, L, M, C = D2 // 2, H * W, B * H * W, R + 2 * N
        dev = xz.device
        shadow.invalidate(dev)                   # a backward pass is under way: the weights are about to change
        isz, xz_bf16 = xz.element_size(), int(xz.dtype == torch.bfloat16)
        if dout.dtype not in (torch.float32, torch.bfloat16):
            dout = dout.float()
        dout = dout.contiguous()
        dxz = torch.empty_like(xz)
        dy = torch.empty((B, L, D), device=dev, dtype=torch.float32)
        du4 = torch.empty((4, B, L, D), device=dev, dtype=torch.float32)
        ddelta = torch.empty_like(du4)
        dproj = arena.zeros_like(proj)
        scratch = _dwconv_bwd_scratch(B, D, H, W, dev)
        sizes = (A.numel(), Dv.numel(), bias.numel(), gamma.numel(), beta.numel(), cw.numel(), cb.numel() if cb is not None else 0)
        zbuf = arena.zeros(sum(sizes), dev)
        dA, dD, dbias, dgamma, dbeta, dcw, dcb = zbuf.split(sizes)
        Q = MsScanBwdParams()
        _ss2d_params(Q.f, xc, proj, delta, A, Dv, bias, None, x_state, H, W, N, R, a_is_log=True)
        if ctx.act:
            Q.f.delta_softplus |= 512               # MS_SCAN_DELTA_ACTIVATED: `delta` holds delta'
        Q.dout_batch_stride, Q.dout_group_stride, Q.dout_d_stride, Q.dout_l_stride = L * D, 0, 1, D
        Q.du_batch_stride, Q.du_group_stride, Q.du_d_stride, Q.du_l_stride = L * D, B * L * D, 1, D
        Q.ddelta_batch_stride, Q.ddelta_group_stride, Q.ddelta_d_stride, Q.ddelta_l_stride = L * D, B * L * D, 1, D
        Q.dB_batch_stride, Q.dB_group_stride, Q.dB_dstate_stride, Q.dB_l_stride = L * 4 * C, C, 1, 4 * C
        Q.dC_batch_stride, Q.dC_group_stride, Q.dC_dstate_stride, Q.dC_l_stride = L * 4 * C, C, 1, 4 * C
        Q.dout, Q.du, Q.ddelta = dy.data_ptr(), du4.data_ptr(), ddelta.data_ptr()
        Q.dA, Q.dD, Q.ddelta_bias = dA.data_ptr(), dD.data_ptr(), dbias.data_ptr()
        Q.dB, Q.dC = dproj.data_ptr() + 4 * R, dproj.data_ptr() + 4 * (R + N)
        stream = _lib.current_stream_ptr(dev)
        with _lib.on_device(dev):
            _lib.check(lib.ms_ln_gate_bwd(ysum.data_ptr(), 0, xz.data_ptr() + D * isz, xz_bf16, D2, gamma.data_ptr(),      # dir_stride 0: the merged sum
                                          beta.data_ptr(), eps, dout.data_ptr(), int(dout.dtype == torch.bfloat16),
                                          dy.data_ptr(), dxz.data_ptr() + D * isz, D2, dgamma.data_ptr(), dbeta.data_ptr(),
                                          M, D, stream), "ms_ln_gate_bwd")
            rc = TIMER.launch("scan_bwd", algorithmic_bytes(B, 4 * D, L, N, 4, True), dev,
                              lambda: lib.ms_selective_scan_bwd(ctypes.byref(Q), stream), B * 4 * D * L * N)
            _lib.check(rc, "ms_selective_scan_bwd[ss2d]")
            dwdt = _dtproj_bwd(ddelta, proj, wdt, dproj, B, L, D, R, C)
            # x_proj backward: input gradient in fp32 straight out of the GEMM, split-K weight gradient
            dpm = dproj.view(M, 4 * C)
            if ctx.mfma and linear_bwd_fused_ok(dpm, xc.view(M, D), wx):
                dxe, dwx = linear_bwd_fused(dpm, xc.view(M, D), wx)    # both products from one pass over dproj and xc (early stages)
            elif ctx.mfma:
                dxe = gemm(dpm, wx, b_trans=True)                      # fp32 dproj read in place, fp32 result
                dwx = weight_grad(dpm, xc.view(M, D))
            elif ctx.f32mm:
                dxe = gemm_f32(dpm, wx.t().contiguous())              # plain / plain against W^T (see gemm_ops._LinearF32)
                dwx = weight_grad_f32(dpm, xc.view(M, D))
            elif xm is not None:
                dpm = dpm.to(xm.dtype)
                dxe = torch.mm(dpm, wx, out_dtype=torch.float32)
            else:
                xm = xc.view(M, D)
                dxe = torch.mm(dpm, wx)
            S = _split_k(M)
            if ctx.mfma or ctx.f32mm:
                pass
            elif S > 1:
                a, b = dpm.view(S, M // S, 4 * C).transpose(1, 2), xm.view(S, M // S, D)
                dwx = (torch.bmm(a, b, out_dtype=torch.float32) if xm.dtype != torch.float32 else torch.bmm(a, b)).sum(dim=0)
            else:
                dwx = torch.mm(dpm.t(), xm).float()
            _lib.check(lib.ms_dwconv3x3_silu_nhwc_bwd(
                xz.data_ptr(), xz_bf16, cw.data_ptr(), cb.data_ptr() if cb is not None else None, du4.data_ptr(), 4, B * L * D,
                dxe.data_ptr(), dxz.data_ptr(), xz_bf16, D2, scratch.data_ptr(), dcw.data_ptr(),
                dcb.data_ptr() if cb is not None else None, B, D, H, W, D2, stream), "ms_dwconv3x3_silu_nhwc_bwd")
        cwd, cbd, wxd, wxs = ctx.dtypes
        return (dxz, dcw.view(cw.shape).to(cwd), (dcb.to(cbd) if cb is not None else None), dwx.view(wxs).to(wxd), dwdt,
                dA.view(A.shape), dD, dbias, dgamma, dbeta, None, None, None, None, None, None)


def ss2d_inner(xz, mod):
    """xz = in_proj(x) (B,H,W,2*d_inner) -> out_norm(merge(scan(conv(x)))) * silu(z), (B,H,W,d_inner), ready for out_proj
    (bf16 under bf16 autocast, else fp32).  `mod` is the SS2D module (parameters)."""
    ac = torch.is_autocast_enabled()
    mm_dtype = torch.get_autocast_dtype("cuda") if ac else None
    if mm_dtype == torch.float32:
        mm_dtype = None
    out_bf16 = mm_dtype == torch.bfloat16
    with torch.autocast(device_type="cuda", enabled=False):
        return _SS2DInner.apply(xz, mod.conv2d.weight, mod.conv2d.bias, mod.x_proj_weight, mod.dt_projs_weight, mod.A_logs,
                                mod.Ds.view(-1), mod.dt_projs_bias.view(-1), mod.out_norm.weight, mod.out_norm.bias,
                                mod.out_norm.eps, mod.d_state, mod.dt_rank, out_bf16, mm_dtype, torch.is_grad_enabled())


def _projections(xc, x_proj_weight, dt_projs_weight, d_state, dt_rank):
    """x_proj follows the ambient autocast (bf16 under autocast, like the reference's einsum would, MedMamba.py:397);
    dt_proj is fp32 always (MedMamba.py:403-409) and runs inside the scan's autograd node (_dtproj_fwd/_dtproj_bwd);
    returns (proj (B,L,4,C), None, Wdt)."""
    B, H, W, D = xc.shape
    C = dt_rank + 2 * d_state
    M = B * H * W
    from .ss2d_ops import linear_splitk
    proj = linear_splitk(xc.view(M, D), x_proj_weight.view(4 * C, D), out_fp32=True)        # (M, 4C) fp32
    return proj.view(B, H * W, 4, C), None, dt_projs_weight


def ss2d_core_norm_gate(xc, z, mod):
    """xc (B,H,W,D) fp32 conv output, z (B,H,W,D) gate (view of xz) -> out_norm(merge(scan)) * silu(z), (B,H,W,D) in the
    dtype out_proj will consume (bf16 under bf16 autocast, else fp32).  `mod` is the SS2D module (parameters)."""
    B, H, W, D = xc.shape
    proj, delta, wdt = _projections(xc, mod.x_proj_weight, mod.dt_projs_weight, mod.d_state, mod.dt_rank)
    out_bf16 = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
    with torch.autocast(device_type="cuda", enabled=False):
        # A = -exp(A_logs) (MedMamba.py:407) is formed inside the scan kernels (MS_SCAN_A_IS_LOG): no exp/neg launches,
        # and the gradient comes back w.r.t. A_logs directly
        return _SS2DScanNormGate.apply(xc, proj, delta, wdt, mod.A_logs.float(), mod.Ds.float().view(-1), mod.dt_projs_bias.float().view(-1),
                                       z, mod.out_norm.weight, mod.out_norm.bias, mod.out_norm.eps, H, W,
                                       mod.d_state, mod.dt_rank, out_bf16)


def ss2d_core(xc, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, d_state, dt_rank):
    """xc (B,H,W,D) fp32 -> merged scan output y (B,H,W,D) fp32 (without the norm/gate tail)."""
    B, H, W, D = xc.shape
    proj, delta, wdt = _projections(xc, x_proj_weight, dt_projs_weight, d_state, dt_rank)
    with torch.autocast(device_type="cuda", enabled=False):
        As = -torch.exp(A_logs.float())
        y = _SS2DScan.apply(xc, proj, delta, wdt, As, Ds.float().view(-1), dt_projs_bias.float().view(-1), H, W,
                            d_state, dt_rank)
    return y.view(B, H, W, D)


# ---- SSD (Mamba-2 form) on the same kernels, in pixel order ----------------------------------------------------------------
class _SSDScanMerge(torch.autograd.Function):
    """4-direction SSD scan of SS2D_with_SSD / CrossMamba (CNN_Mamba.py:494-552) on the conv output in PIXEL order -- no
    gathered copies of the four scan orders and no inverse gathers: `xc` (B,H,W,conv_dim) = [x (Ds) | B (N) | C (N) | dt
    (nheads)] per pixel, ngroups == 1.  Heads = (direction, head); every head's state is the concatenation of the four
    directions' B/C (CNN_Mamba.py:506-519), so the state axis is covered by 4 * ceil(N/16) launches: launch (j, slice) takes
    16 states of B/C through direction j's pixel order for ALL groups (MS_SCAN_BC_MAP) while u / dt / y follow each group's own direction, and
    adds into the same y (MS_SCAN_ACCUMULATE).  Scalar decay per head (A stride 0 over states)."""

    @staticmethod
    def forward(ctx, xc, As, Dsv, dt_bias, H, W, Ds, N, nheads, headdim, d_has_hdim):
        _lib.require_cuda(xc, As, Dsv, dt_bias)
        lib = _lib.lib()
        B, conv = xc.shape[0], xc.shape[-1]
        L = H * W
        xc = xc.contiguous().float()
        # per-channel views of the per-head parameters (channel = (direction, head, p))
        A_col = As.detach().float().view(4 * nheads, 1).expand(4 * nheads, headdim).reshape(4 * Ds, 1).contiguous()
        D_full = (Dsv.detach().float().reshape(4 * Ds) if d_has_hdim
                  else Dsv.detach().float().view(4 * nheads, 1).expand(4 * nheads, headdim).reshape(4 * Ds)).contiguous()
        bias_full = dt_bias.detach().float().view(4 * nheads, 1).expand(4 * nheads, headdim).reshape(4 * Ds).contiguous()
        delta = xc[..., Ds + 2 * N:].repeat_interleave(headdim, dim=-1).view(B, L, Ds)      # dt of a pixel, per channel
        y4 = torch.empty((4, B, L, Ds), device=xc.device, dtype=torch.float32)
        n_chunks = lib.ms_scan_n_chunks(L)
        slices = _ssd_slices(N)                          # (direction j, first state, states) per launch
        x_state = torch.empty((len(slices), B, n_chunks, _SSD_SLICE, 4 * Ds), device=xc.device, dtype=torch.float32)
        stream = _lib.current_stream_ptr(xc.device)
        with _lib.on_device(xc.device):
            if N == _SSD_SLICE and SSD_ONE_LAUNCH_FWD:
                # all four directions' B/C slices in ONE launch (MS_SCAN_BC_MAP(4)): one pass over u / delta / y instead of
                # four, the saved states in the slice-major layout the four backward launches read
                P = MsScanParams()
                _ssd_params(P, xc, delta, A_col, D_full, bias_full, y4, x_state, B, L, H, W, Ds, N, conv, 4, 0, 4 * N, False)
                rc = TIMER.launch("scan_fwd", algorithmic_bytes(B, 4 * Ds, L, 4 * N, 4, False), xc.device,
                                  lambda: lib.ms_selective_scan_fwd(ctypes.byref(P), stream))
                _lib.check(rc, "ms_selective_scan_fwd[ssd, all directions]")
            else:
                for i, (j, s0, ns) in enumerate(slices):
                    P = MsScanParams()
                    _ssd_params(P, xc, delta, A_col, D_full if i == 0 else None, bias_full, y4, x_state[i], B, L, H, W, Ds, N, conv,
                                j, s0, ns, i > 0)
                    rc = TIMER.launch("scan_fwd", algorithmic_bytes(B, 4 * Ds, L, ns, 4, False), xc.device,
                                      lambda: lib.ms_selective_scan_fwd(ctypes.byref(P), stream))
                    _lib.check(rc, "ms_selective_scan_fwd[ssd]")
        ctx.save_for_backward(xc, delta, A_col, D_full, bias_full, x_state)
        ctx.geom = (H, W, Ds, N, nheads, headdim, bool(d_has_hdim), As.shape, Dsv.shape, dt_bias.shape)
        return ((y4[0] + y4[2]) + y4[1]) + y4[3]

    @staticmethod
    def backward(ctx, dy):
        xc, delta, A_col, D_full, bias_full, x_state = ctx.saved_tensors
        H, W, Ds, N, nheads, headdim, d_has_hdim, a_shape, d_shape, b_shape = ctx.geom
        lib = _lib.lib()
        B, conv, L = xc.shape[0], xc.shape[-1], H * W
        dy = dy.contiguous().float()
        du4 = torch.empty((4, B, L, Ds), device=xc.device, dtype=torch.float32)
        dd4 = torch.empty_like(du4)
        dxc = torch.zeros_like(xc)                       # the B|C columns are accumulated by the kernels (atomics)
        # dA comes back dense, (channel, state of the launch): one accumulator per distinct slice width
        widths = sorted({ns for _, _, ns in _ssd_slices(N)})
        sizes = tuple(4 * Ds * w for w in widths) + (4 * Ds, 4 * Ds)
        zbuf = arena.zeros(sum(sizes), xc.device)
        *dA_w, dD, dbias = zbuf.split(sizes)
        dA_of = dict(zip(widths, dA_w))
        stream = _lib.current_stream_ptr(xc.device)
        # all four direction slices in ONE launch (MS_SCAN_BC_MAP(4), csrc/scan_bwd_ssd.hip) when the forward ran that way too
        # (same slice-major saved states); otherwise one launch per slice, accumulating
        one = N == _SSD_SLICE and SSD_ONE_LAUNCH_FWD and SSD_ONE_LAUNCH_BWD
        launches = [(4, 0, 4 * N)] if one else _ssd_slices(N)
        with _lib.on_device(xc.device):
            for i, (j, s0, ns) in enumerate(launches):
                Q = MsScanBwdParams()
                _ssd_params(Q.f, xc, delta, A_col, D_full if i == 0 else None, bias_full, None, x_state if one else x_state[i], B, L, H, W,
                            Ds, N, conv, j, s0, ns, i > 0)
                Q.dout_batch_stride, Q.dout_group_stride, Q.dout_d_stride, Q.dout_l_stride = L * Ds, 0, 1, Ds
                Q.du_batch_stride, Q.du_group_stride, Q.du_d_stride, Q.du_l_stride = L * Ds, B * L * Ds, 1, Ds
                Q.ddelta_batch_stride, Q.ddelta_group_stride, Q.ddelta_d_stride, Q.ddelta_l_stride = L * Ds, B * L * Ds, 1, Ds
                Q.dB_batch_stride, Q.dB_group_stride, Q.dB_dstate_stride, Q.dB_l_stride = L * conv, 0, 1, conv
                Q.dC_batch_stride, Q.dC_group_stride, Q.dC_dstate_stride, Q.dC_l_stride = L * conv, 0, 1, conv
                Q.dout, Q.du, Q.ddelta = dy.data_ptr(), du4.data_ptr(), dd4.data_ptr()
                Q.dA, Q.ddelta_bias = dA_of[N if one else ns].data_ptr(), dbias.data_ptr()
                Q.dD = dD.data_ptr() if i == 0 else None
                Q.dB, Q.dC = dxc.data_ptr() + 4 * (Ds + s0), dxc.data_ptr() + 4 * (Ds + N + s0)
                rc = TIMER.launch("scan_bwd", algorithmic_bytes(B, 4 * Ds, L, ns, 4, True), xc.device,
                                  lambda: lib.ms_selective_scan_bwd(ctypes.byref(Q), stream))
                _lib.check(rc, "ms_selective_scan_bwd[ssd]")
        dxc[..., :Ds] = du4.sum(dim=0).view(B, H, W, Ds)
        dxc[..., Ds + 2 * N:] = dd4.sum(dim=0).view(B, H, W, nheads, headdim).sum(dim=-1)
        dAs = sum(dA_of[w].view(4 * nheads, headdim * w).sum(dim=1) for w in widths).view(a_shape)
        dDs = dD.view(d_shape) if d_has_hdim else dD.view(4 * nheads, headdim).sum(dim=1).view(d_shape)
        dbt = dbias.view(4 * nheads, headdim).sum(dim=1).view(b_shape)
        return dxc, dAs, dDs, dbt, None, None, None, None, None, None, None


_SSD_SLICE = 16      # states per backward launch (the backward kernels keep <= 16 states of a channel in registers)
SSD_ONE_LAUNCH_FWD = os.environ.get("MEDSCAN_SSD_ONE_LAUNCH_FWD", "1") == "1"
SSD_ONE_LAUNCH_BWD = os.environ.get("MEDSCAN_SSD_ONE_LAUNCH_BWD", "1") == "1"


def _ssd_slices(N):
    return [(j, s0, min(_SSD_SLICE, N - s0)) for j in range(4) for s0 in range(0, N, _SSD_SLICE)]


def _ssd_params(P, xc, delta, A_col, D_full, bias_full, y4, x_state, B, L, H, W, Ds, N, conv, j, s0, ns, accumulate):
    """One slice of the SSD state axis: states s0..s0+ns of direction j's B/C, read through direction j's pixel order
    (see _SSDScanMerge)."""
    P.batch, P.dim, P.seqlen, P.dstate, P.n_groups = B, 4 * Ds, L, ns, 4
    P.delta_softplus = 1 | ((j + 1) << 4) | (4 if accumulate else 0)   # SOFTPLUS | BC_MAP(j) | ACCUMULATE
    P.map_h, P.map_w = H, W
    P.u_batch_stride, P.u_group_stride, P.u_d_stride, P.u_l_stride = L * conv, 0, 1, conv
    P.delta_batch_stride, P.delta_group_stride, P.delta_d_stride, P.delta_l_stride = L * Ds, 0, 1, Ds
    P.out_batch_stride, P.out_group_stride, P.out_d_stride, P.out_l_stride = L * Ds, B * L * Ds, 1, Ds
    P.A_d_stride, P.A_dstate_stride = 1, 0
    P.B_batch_stride, P.B_group_stride, P.B_dstate_stride, P.B_l_stride = L * conv, 0, 1, conv
    P.C_batch_stride, P.C_group_stride, P.C_dstate_stride, P.C_l_stride = L * conv, 0, 1, conv
    P.u, P.delta, P.A = xc.data_ptr(), delta.data_ptr(), A_col.data_ptr()
    P.B, P.C = xc.data_ptr() + 4 * (Ds + s0), xc.data_ptr() + 4 * (Ds + N + s0)
    P.D = D_full.data_ptr() if D_full is not None else None
    P.delta_bias = bias_full.data_ptr()
    P.out = y4.data_ptr() if y4 is not None else None
    P.x = x_state.data_ptr() if x_state is not None else None


def ssd_scan_merge_pixel(xc, As, Dsv, dt_bias, d_ssm, d_state, nheads, headdim, d_has_hdim):
    """(B,H,W,conv_dim) fp32 -> (B,H,W,d_ssm) fp32: the four directions scanned and merged, for ngroups == 1."""
    B, H, W, _ = xc.shape
    with torch.autocast(device_type="cuda", enabled=False):
        y = _SSDScanMerge.apply(xc.float(), As, Dsv, dt_bias, H, W, d_ssm, d_state, nheads, headdim, d_has_hdim)
    return y.view(B, H, W, d_ssm)
