"""SSD (Mamba-2) variant of the model -- module surface of the reference's CNN_Mamba.py, the file its train.py /
ddp_train.py import as shipped (`from CNN_Mamba import VSSM as medmamba`, train.py:11): ConvTConvPW :43,
SS2D_with_SSD :322 (named MedSSD in CrossMamba/CrossMamba_fusion_2b2.py:390), SS_Conv_SSD :583, VSSLayer :622,
VSSM :752.  Same constructor arguments, forward signatures and state_dict keys.

The arithmetic of this path lives in a third-party package that is NOT in the reference tree:
`mamba_ssm==2.2.2` (README.md:7) -- `ops.triton.ssd_combined.mamba_chunk_scan_combined` and
`ops.triton.layernorm_gated.RMSNorm`.  They are restated here from the call-site contract (CNN_Mamba.py:506-537,
428-431,555-556) and the published recurrence
    h_t = exp(dt_t*A_h) h_{t-1} + dt_t * B_t (x) x_t ,   y_t = C_t . h_t + D_h x_t ,   dt = softplus(dt + dt_bias)
PARITY UNPINNED against the Triton kernels (nothing in the reference pins results at this boundary);
the restatement is cross-checked against the pinned S6 oracle by expanding scalar-A heads to diagonal A
(tests/test_ssd_cpu.py, tests/test_ssd_gpu.py).

Mapping onto the gfx950 S6 kernels: a head of `headdim` channels = `headdim` channels sharing A and dt; the state axis
(ngroups*d_state, = 4*d_state in SS2D_with_SSD where the four directions' B/C are concatenated, CNN_Mamba.py:506-519)
is split into slices of <= 16 states whose outputs add (the recurrence is independent per state), one
`selective_scan_fn` call per slice.  No CPU fallback.
"""
import math
import os
from functools import partial
from typing import Callable

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

from . import _lib
from . import medmamba as mm
from .block_ops import block_tail, conv_branch, layernorm_rows, split_halves
from .medmamba import CONV_CHANNELS_LAST, DropPath, PatchEmbed2D, PatchMerging2D, channel_shuffle
from .selective_scan_interface import selective_scan_fn
from .ss2d_fused import dwconv3x3_silu_nhwc

SSD_PIXEL_ORDER = os.environ.get("MEDSCAN_SSD_PIXEL", "1") == "1"     # ssd_scan_merge: pixel-order kernels vs gathered copies
_STATE_SLICE = 16      # states per kernel call (the backward kernels are built for dstate <= 16)


# SSD as chunked matrix products (the "state space duality" form of Mamba-2, which the reference's dependency implements in
# Triton): one launch set for ALL states instead of one scan launch per 16 states.  Used when the state is wide enough
# for the GEMMs to win: measured at 512 states (VFEFM, bs 32) 2.34 s -> 1.36 s per step; at 64 states (CNN_Mamba.VSSM,
# bs 32) the scan kernels win, 46 vs 74 ms.  0 = always the scan kernels.
# Round 3: with the chunked form on this package's own MFMA kernels (csrc/ssd_chunk.hip, _SSDChunkKernels below) it wins from 64 states
# on as well (CNN_Mamba.VSSM bs 32: see DESIGN.md 9): the default threshold is 64.
SSD_CHUNKED_MIN_STATE = int(os.environ.get("MEDSCAN_SSD_CHUNKED_MIN_STATE", "64"))
# Scans whose chunk-state tensor is at most this large keep their intermediates for backward (plain autograd); larger ones
# keep only their operands and recompute (_SSDChunked).  Measured on VFEFM bs 32 (ms/step, peak HBM): 0 GB 883 / 52 GiB,
# 0.5 GB 861 / 58, 1 GB 816 / 97, 2 GB 798 / 141, everything kept 1056* / 250 (*before the carry kernel).  288 GB of HBM3E
# is there to be used: 2 GB.
SSD_KEEP_STATE_BYTES = int(float(os.environ.get("MEDSCAN_SSD_KEEP_STATE_GB", "2")) * 2 ** 30)
_SSD_CHUNK = int(os.environ.get("MEDSCAN_SSD_CHUNK", "64"))      # positions per chunk (64 / 128 / 256 measured within 8 % of each other on VFEFM)


class _ChunkCarry(torch.autograd.Function):
    """S_in[z] = decay[z-1] * S_in[z-1] + S[z-1] over the chunks (ms_ssd_chunk_carry), in the layout the state GEMMs use:
    S (b, c, g, n, hg, p), decay (b, c, g*hg).  Backward = the same sweep run from the last chunk to the first."""

    @staticmethod
    def forward(ctx, S, decay):
        _lib.require_cuda(S, decay)
        b, c, g, n, hg, p = S.shape
        S, decay = S.contiguous(), decay.contiguous()
        out = torch.empty_like(S)
        with _lib.on_device(S.device):
            _lib.check(_lib.lib().ms_ssd_chunk_carry(S.data_ptr(), decay.data_ptr(), out.data_ptr(), None, None, b, c, g, n, hg, p, 0,
                                                     _lib.current_stream_ptr(S.device)), "ms_ssd_chunk_carry")
        ctx.save_for_backward(out, decay)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, decay = ctx.saved_tensors
        b, c, g, n, hg, p = out.shape
        dout = dout.contiguous()
        dS = torch.empty_like(out)
        ddecay = torch.zeros_like(decay)               # d/d decay[z] = <gradient reaching S[z], S_in[z]>, reduced in the kernel
        with _lib.on_device(out.device):
            _lib.check(_lib.lib().ms_ssd_chunk_carry(dout.data_ptr(), decay.data_ptr(), dS.data_ptr(), out.data_ptr(),
                                                     ddecay.data_ptr(), b, c, g, n, hg, p, 1,
                                                     _lib.current_stream_ptr(out.device)), "ms_ssd_chunk_carry[reverse]")
        return dS, ddecay


SSD_CARRY_KERNEL = os.environ.get("MEDSCAN_SSD_CARRY_KERNEL", "1") == "1"


def _ssd_chunked(x, dt, A, B, C, D, dt_bias, dt_softplus, Q=_SSD_CHUNK):
    """h_t = exp(dt_t A) h_{t-1} + dt_t B_t (x) x_t,  y_t = C_t . h_t + D x_t, evaluated chunk-wise:
    inside a chunk   Y = ((C B^T) o L) X'          L[i,j] = exp(sum_{j<k<=i} dt_k A), i >= j   (X' = dt * x)
    chunk states     S_c = sum_j exp(sum_{k>j} dt_k A) B_j^T X'_j ;  carried across chunks by their total decays
    state -> output  Y += exp(sum_{k<=i} dt_k A) C_i S_{c-1}
    x (b,l,h,p), dt (b,l,h), A (h), B/C (b,l,g,n) -> (b,l,h,p) fp32.  All products are batched GEMMs (hipBLASLt) in fp32;
    autograd differentiates them."""
    b, l, h, p = x.shape
    g, n = B.shape[2], B.shape[3]
    hg = h // g
    dt = dt.float()
    if dt_bias is not None:
        dt = dt + dt_bias.float()
    if dt_softplus:
        dt = F.softplus(dt)
    x = x.float()
    nc = (l + Q - 1) // Q
    pad = nc * Q - l
    a = dt * A.float()                                             # (b,l,h) log-decay per step, <= 0
    xd = x * dt.unsqueeze(-1)
    Bf, Cf = B.float(), C.float()
    if pad:                                                         # dt = 0 steps: decay 1, no input; their outputs are dropped
        a, xd, Bf, Cf = (F.pad(t, (0, 0) * (t.dim() - 2) + (0, pad)) for t in (a, xd, Bf, Cf))
    a = a.reshape(b, nc, Q, h).permute(0, 1, 3, 2)                  # (b,c,h,Q)
    cum = torch.cumsum(a, dim=-1)
    xd = xd.reshape(b, nc, Q, g, hg * p)
    Bc = Bf.reshape(b, nc, Q, g, n).permute(0, 1, 3, 2, 4)          # (b,c,g,Q,n)
    Cc = Cf.reshape(b, nc, Q, g, n).permute(0, 1, 3, 2, 4)
    # inside the chunk
    seg = cum.unsqueeze(-1) - cum.unsqueeze(-2)                     # (b,c,h,Q,Q): sum_{j<k<=i}
    tri = torch.ones(Q, Q, device=x.device, dtype=torch.bool).tril()
    Lm = torch.exp(seg.masked_fill(~tri, float("-inf")))
    CB = torch.matmul(Cc, Bc.transpose(-1, -2))                     # (b,c,g,Q,Q)
    M = Lm.view(b, nc, g, hg, Q, Q) * CB.unsqueeze(3)
    Xh = xd.view(b, nc, Q, g, hg, p).permute(0, 1, 3, 4, 2, 5)      # (b,c,g,hg,Q,p)
    y = torch.matmul(M, Xh)                                         # (b,c,g,hg,Q,p)
    # chunk states (per head: n x p), the decay of input j up to the chunk's end
    dec_in = torch.exp(cum[..., -1:] - cum)                         # (b,c,h,Q)
    Xs = (Xh * dec_in.view(b, nc, g, hg, Q, 1)).permute(0, 1, 2, 4, 3, 5).reshape(b, nc, g, Q, hg * p)
    S = torch.matmul(Bc.transpose(-1, -2), Xs)                      # (b,c,g,n,hg*p)
    if SSD_CARRY_KERNEL and x.is_cuda and p % 4 == 0:
        # carry across chunks: S_in[z] = exp(tot[z-1]) S_in[z-1] + S[z-1] (tot = the chunks' total log-decays, per head): one
        # sweep over the chunks in the GEMMs' own layout
        S_in = _ChunkCarry.apply(S.view(b, nc, g, n, hg, p), torch.exp(cum[..., -1])).view(b, nc, g, n, hg * p)
    else:
        # carry across chunks: S_in[c] = sum_{c' < c} exp(sum_{c' < k < c} tot_k) S[c'],  tot = the chunks' total log-decays
        tot = F.pad(cum[..., -1], (0, 0, 1, 0)).permute(0, 2, 1)       # (b,h,1+c): a leading zero for the (zero) initial state
        ct = torch.cumsum(tot, dim=-1)
        segc = ct.unsqueeze(-1) - ct.unsqueeze(-2)                      # (b,h,1+c,1+c)
        tric = torch.ones(nc + 1, nc + 1, device=x.device, dtype=torch.bool).tril()
        # Dc[b,h,z,c'] = exp(ct[z] - ct[c'+1]) for c' < z: what is left of chunk c''s end state when chunk z starts; 0 otherwise
        Dc = torch.exp(segc.masked_fill(~tric, float("-inf")))[:, :, :-1, 1:]
        Sh = S.view(b, nc, g, n, hg, p).permute(0, 2, 4, 1, 3, 5).reshape(b, h, nc, n * p)      # (b,h,c,n*p)
        S_in = torch.matmul(Dc, Sh).view(b, g, hg, nc, n, p)            # (b,g,hg,z,n,p)
        S_in = S_in.permute(0, 3, 1, 4, 2, 5).reshape(b, nc, g, n, hg * p)
    y_off = torch.matmul(Cc, S_in).view(b, nc, g, Q, hg, p).permute(0, 1, 2, 4, 3, 5)      # (b,c,g,hg,Q,p)
    y = y + y_off * torch.exp(cum).view(b, nc, g, hg, Q, 1)
    y = y.permute(0, 1, 4, 2, 3, 5).reshape(b, nc * Q, h, p)[:, :l]
    if D is not None:
        Df = D.float()
        y = y + x * (Df.view(h, p) if Df.dim() == 2 else Df.view(h, 1))
    return y


class _SSDChunked(torch.autograd.Function):
    """_ssd_chunked with its intermediates recomputed in backward: the per-chunk states are (n x h*p) per (batch, chunk) --
    1.6 GB per tensor for one stage-0 block of VFEFM at batch 32 -- so only the operands are kept (one extra forward per scan
    in backward; measured on VFEFM bs 32: peak 250 -> see DESIGN.md section 9)."""

    @staticmethod
    def forward(ctx, x, dt, A, B, C, D, dt_bias, dt_softplus):
        ctx.save_for_backward(x, dt, A, B, C, D, dt_bias)
        ctx.dt_softplus = dt_softplus
        with torch.no_grad():
            return _ssd_chunked(x, dt, A, B, C, D, dt_bias, dt_softplus)

    @staticmethod
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        ins = [t.detach().requires_grad_(ctx.needs_input_grad[i]) if t is not None else None for i, t in enumerate(saved)]
        with torch.enable_grad():
            y = _ssd_chunked(*ins, ctx.dt_softplus)
        wanted = [t for t in ins if t is not None and t.requires_grad]
        grads = iter(torch.autograd.grad(y, wanted, dy))
        return tuple(next(grads) if (t is not None and t.requires_grad) else None for t in ins) + (None,)


# ---- the chunked form on this package's MFMA kernels (csrc/ssd_chunk.hip) ------------------------------------------------------
SSD_CHUNK_KERNELS = os.environ.get("MEDSCAN_SSD_CHUNK_KERNELS", "1") == "1"


def _ssd_kernels_ok(x, B):
    """shapes the fused chunk kernels take: headdim 64, one B/C group, a state dimension that is a multiple of 64"""
    return SSD_CHUNK_KERNELS and x.is_cuda and x.shape[-1] == 64 and B.shape[2] == 1 and B.shape[3] % 64 == 0


def _ssd_chunk_forward(x, dt, A, B, C, D, dt_bias, dt_softplus):
    """ms_ssd_chunk_fwd -> ms_ssd_chunk_carry -> ms_ssd_chunk_fwd_off.  Returns y and the workspaces the backward re-uses."""
    b, l, h, p = x.shape
    n = B.shape[3]
    nc = (l + 63) // 64
    dev = x.device
    f32 = lambda t: t.detach().float().contiguous()
    x, dt, A, B, C = f32(x), f32(dt), f32(A), f32(B).view(b, l, n), f32(C).view(b, l, n)
    D = f32(D) if D is not None else None
    bias = f32(dt_bias) if dt_bias is not None else None
    dtv = torch.empty((b, nc, h, 64), device=dev, dtype=torch.float32)
    cum = torch.empty_like(dtv)
    decay = torch.empty((b, nc, h), device=dev, dtype=torch.float32)
    CB = torch.empty((b, nc, 64, 64), device=dev, dtype=torch.float32)
    S = torch.empty((b, nc, n, h * p), device=dev, dtype=torch.float32)
    y = torch.empty((b, l, h, p), device=dev, dtype=torch.float32)
    lib, st = _lib.lib(), _lib.current_stream_ptr(dev)
    with _lib.on_device(dev):
        _lib.check(lib.ms_ssd_chunk_fwd(x.data_ptr(), dt.data_ptr(), A.data_ptr(), B.data_ptr(), C.data_ptr(),
                                        bias.data_ptr() if bias is not None else None, int(bool(dt_softplus)), dtv.data_ptr(), cum.data_ptr(),
                                        decay.data_ptr(), CB.data_ptr(), S.data_ptr(), y.data_ptr(), b, l, h, p, n, st), "ms_ssd_chunk_fwd")
        S_in = torch.empty_like(S)
        _lib.check(lib.ms_ssd_chunk_carry(S.data_ptr(), decay.data_ptr(), S_in.data_ptr(), None, None, b, nc, 1, n, h, p, 0, st),
                   "ms_ssd_chunk_carry")
        _lib.check(lib.ms_ssd_chunk_fwd_off(x.data_ptr(), C.data_ptr(), S_in.data_ptr(), cum.data_ptr(),
                                            D.data_ptr() if D is not None else None, int(D is not None and D.dim() == 2), y.data_ptr(),
                                            b, l, h, p, n, st), "ms_ssd_chunk_fwd_off")
    return y, (x, dt, A, B, C, D, bias, dtv, cum, decay, CB, S_in)


class _SSDChunkKernels(torch.autograd.Function):
    """mamba_chunk_scan_combined on csrc/ssd_chunk.hip: forward and backward on the exact-fp32 matrix instruction, no torch.matmul, none
    of the (chunks x heads x 64 x 64) mask tensors.  The entering states S_in ((l / 64) x n x h*p floats per sample -- 1.6 GB for one
    stage-0 scan of VFEFM at batch 32) are kept for the backward up to MEDSCAN_SSD_KEEP_STATE_GB and recomputed above it."""

    @staticmethod
    def forward(ctx, x, dt, A, B, C, D, dt_bias, dt_softplus):
        y, ws = _ssd_chunk_forward(x, dt, A, B, C, D, dt_bias, dt_softplus)
        xf, dtf, Af, Bf, Cf, Df, bias, dtv, cum, decay, CB, S_in = ws
        keep = S_in.numel() * 4 <= SSD_KEEP_STATE_BYTES
        ctx.save_for_backward(xf, dtf, Af, Bf, Cf, Df, bias, dtv, cum, decay, CB, S_in if keep else None)
        ctx.dt_softplus = bool(dt_softplus)
        ctx.meta = (x.dtype, dt.dtype, A.dtype, B.dtype, C.dtype, D.dtype if D is not None else None, dt_bias.dtype if dt_bias is not None else None,
                    tuple(B.shape), tuple(D.shape) if D is not None else None, tuple(dt_bias.shape) if dt_bias is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, dt, A, B, C, D, bias, dtv, cum, decay, CB, S_in = ctx.saved_tensors
        b, l, h, p = x.shape
        n = B.shape[2]
        nc = (l + 63) // 64
        dev = x.device
        dy = dy.float().contiguous()
        lib, st = _lib.lib(), _lib.current_stream_ptr(dev)
        with _lib.on_device(dev):
            S = torch.empty((b, nc, n, h * p), device=dev, dtype=torch.float32)
            if S_in is None:               # recompute the entering states (states only: no Y)
                _lib.check(lib.ms_ssd_chunk_fwd(x.data_ptr(), dt.data_ptr(), A.data_ptr(), B.data_ptr(), C.data_ptr(),
                                                bias.data_ptr() if bias is not None else None, int(ctx.dt_softplus), dtv.data_ptr(),
                                                cum.data_ptr(), decay.data_ptr(), CB.data_ptr(), S.data_ptr(), None, b, l, h, p, n, st),
                           "ms_ssd_chunk_fwd[states]")
                S_in = torch.empty_like(S)
                _lib.check(lib.ms_ssd_chunk_carry(S.data_ptr(), decay.data_ptr(), S_in.data_ptr(), None, None, b, nc, 1, n, h, p, 0, st),
                           "ms_ssd_chunk_carry")
            dS_in, dcum = S, torch.empty((b, nc, h, 64), device=dev, dtype=torch.float32)          # (S is free again: dS_in takes its place)
            _lib.check(lib.ms_ssd_chunk_bwd_off(dy.data_ptr(), C.data_ptr(), S_in.data_ptr(), cum.data_ptr(), dS_in.data_ptr(), dcum.data_ptr(),
                                                b, l, h, p, n, st), "ms_ssd_chunk_bwd_off")
            dS = torch.empty_like(S_in)
            nd = D.numel() if D is not None else 0
            small = torch.zeros(b * nc * h + 2 * h + nd, device=dev, dtype=torch.float32)          # ddecay | dA | dbias | dD: one fill
            ddecay, dA, dbias = small[:b * nc * h], small[b * nc * h:b * nc * h + h], small[b * nc * h + h:b * nc * h + 2 * h]
            dD = small[b * nc * h + 2 * h:].view(D.shape) if D is not None else None
            dCB = torch.zeros((b, nc, 64, 64), device=dev, dtype=torch.float32)
            if nc > 1:
                dS_in[:, 0].zero_()        # never written by the off kernel (chunk 0 has no entering state) and never read by the carry: keep it finite
            _lib.check(lib.ms_ssd_chunk_carry(dS_in.data_ptr(), decay.data_ptr(), dS.data_ptr(), S_in.data_ptr(), ddecay.data_ptr(),
                                              b, nc, 1, n, h, p, 1, st), "ms_ssd_chunk_carry[reverse]")
            dx = torch.empty_like(x)
            ddt = torch.empty_like(dt)
            dB, dC = torch.empty_like(B), torch.empty_like(C)
            _lib.check(lib.ms_ssd_chunk_bwd(x.data_ptr(), dy.data_ptr(), B.data_ptr(), C.data_ptr(), CB.data_ptr(), S_in.data_ptr(), dS.data_ptr(),
                                            dtv.data_ptr(), cum.data_ptr(), decay.data_ptr(), ddecay.data_ptr(), A.data_ptr(),
                                            D.data_ptr() if D is not None else None, int(D is not None and D.dim() == 2), int(ctx.dt_softplus),
                                            dcum.data_ptr(), dx.data_ptr(), ddt.data_ptr(), dA.data_ptr(),
                                            dbias.data_ptr() if bias is not None else None, dD.data_ptr() if dD is not None else None,
                                            dCB.data_ptr(), dB.data_ptr(), dC.data_ptr(), b, l, h, p, n, st), "ms_ssd_chunk_bwd")
        xd, dtd, Ad, Bd, Cd, Dd, bd, Bshape, Dshape, bshape = ctx.meta
        return (dx.to(xd), ddt.to(dtd), dA.to(Ad), dB.view(Bshape).to(Bd), dC.view(Bshape).to(Cd),
                dD.view(Dshape).to(Dd) if D is not None else None, dbias.view(bshape).to(bd) if bias is not None else None, None)


def mamba_chunk_scan_combined(x, dt, A, B, C, chunk_size=256, D=None, z=None, dt_bias=None, initial_states=None,
                              seq_idx=None, cu_seqlens=None, dt_softplus=False, dt_limit=(0.0, float("inf")),
                              return_final_states=False):
    """Call-site contract of mamba_ssm 2.2.2's SSD scan as used at CNN_Mamba.py:523-537.
    x (b,l,h,p), dt (b,l,h), A (h), B/C (b,l,g,n), D (h) or (h,p), dt_bias (h) -> y (b,l,h,p) in x's dtype.
    `chunk_size` is an algorithmic detail of the Triton kernels, not semantics: ignored."""
    if z is not None or initial_states is not None or seq_idx is not None or cu_seqlens is not None or return_final_states:
        raise RuntimeError("mamba_chunk_scan_combined: z / initial_states / seq_idx / cu_seqlens / final states are not "
                           "used by this repository's models and are not built")
    if dt_limit != (0.0, float("inf")):
        raise RuntimeError("mamba_chunk_scan_combined: dt_limit is not built")
    _lib.require_cuda(x, dt, A, B, C)
    b, l, h, p = x.shape
    g, n = B.shape[2], B.shape[3]
    if h % g != 0:
        raise RuntimeError("mamba_chunk_scan_combined: nheads must be a multiple of ngroups")
    if 0 < SSD_CHUNKED_MIN_STATE <= n and _ssd_kernels_ok(x, B):
        with torch.autocast(device_type="cuda", enabled=False):          # the chunked form on csrc/ssd_chunk.hip (exact-fp32 MFMA)
            return _SSDChunkKernels.apply(x, dt, A, B, C, D, dt_bias, dt_softplus).to(x.dtype)
    if 0 < max(SSD_CHUNKED_MIN_STATE, 128) <= n:       # shapes the kernels do not take: the torch formulation (library GEMMs), wide states only
        with torch.autocast(device_type="cuda", enabled=False):
            # the chunk-state tensors ((l / chunk) x n x h*p floats per sample; autograd would keep ~5 of them) decide whether
            # the scan keeps its intermediates or only its operands
            state_bytes = 4 * b * ((l + _SSD_CHUNK - 1) // _SSD_CHUNK) * n * h * p
            if state_bytes <= SSD_KEEP_STATE_BYTES:
                return _ssd_chunked(x, dt, A, B, C, D, dt_bias, dt_softplus).to(x.dtype)
            return _SSDChunked.apply(x, dt, A, B, C, D, dt_bias, dt_softplus).to(x.dtype)
    dim = h * p
    u = x.float().reshape(b, l, dim).transpose(1, 2)                       # (b, dim, l), channel-last strides
    delta = dt.float().unsqueeze(-1).expand(b, l, h, p).reshape(b, l, dim).transpose(1, 2)
    A_full = A.float().view(h, 1, 1).expand(h, p, 1).reshape(dim, 1)      # same A for every channel and state of a head
    if D is not None:
        Dc = D.float()
        Dc = Dc.reshape(dim) if Dc.dim() == 2 else Dc.view(h, 1).expand(h, p).reshape(dim)
    else:
        Dc = None
    bias = dt_bias.float().view(h, 1).expand(h, p).reshape(dim).contiguous() if dt_bias is not None else None
    Bt = B.float().permute(0, 2, 3, 1)                                      # (b, g, n, l)
    Ct = C.float().permute(0, 2, 3, 1)
    y = None
    for s0 in range(0, n, _STATE_SLICE):
        s1 = min(n, s0 + _STATE_SLICE)
        yi = selective_scan_fn(u, delta, A_full.expand(dim, s1 - s0), Bt[:, :, s0:s1], Ct[:, :, s0:s1],     # stride-0 states: scalar-decay kernels
                               Dc if s0 == 0 else None, None, bias, dt_softplus)
        y = yi if y is None else y + yi
    return y.transpose(1, 2).reshape(b, l, h, p).to(x.dtype)


def _rows_view(t, D):
    """(..., D) tensor (possibly a channel slice of a wider tensor) -> (tensor, row stride) usable by the row kernels."""
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    ps = t.stride(-2) if t.dim() >= 2 else D
    ok = t.stride(-1) == 1 and ps >= D
    exp = ps
    for k in range(t.dim() - 2, -1, -1):                 # leading dims must collapse onto one uniform row stride
        ok = ok and (t.shape[k] == 1 or t.stride(k) == exp)
        exp *= t.shape[k]
    if not ok:
        t = t.contiguous(); ps = D
    return t, ps


class _RMSNormGate(torch.autograd.Function):
    """rmsnorm(y * silu(z)) * weight as one kernel each way (ms_rms_gate_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, y, z, weight, eps, out_dtype):
        _lib.require_cuda(y, z, weight)
        D = y.shape[-1]
        y = y.float().contiguous()
        z, zps = _rows_view(z, D)
        w = weight.detach().float().contiguous()
        npix = y.numel() // D
        out = torch.empty(y.shape, device=y.device, dtype=out_dtype)
        with _lib.on_device(y.device):
            _lib.check(_lib.lib().ms_rms_gate_fwd(y.data_ptr(), 0, 1, z.data_ptr(), int(z.dtype == torch.bfloat16), zps, w.data_ptr(),
                                                  float(eps), out.data_ptr(), int(out_dtype == torch.bfloat16), npix, D,
                                                  _lib.current_stream_ptr(y.device)), "ms_rms_gate_fwd")
        ctx.save_for_backward(y, z, w)
        ctx.geom = (float(eps), zps, weight.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, z, w = ctx.saved_tensors
        eps, zps, wdtype = ctx.geom
        D = y.shape[-1]
        npix = y.numel() // D
        if dout.dtype not in (torch.float32, torch.bfloat16):
            dout = dout.float()
        dout = dout.contiguous()
        dy = torch.empty_like(y)
        dz = torch.empty(y.shape, device=y.device, dtype=z.dtype)
        dw = torch.zeros_like(w)
        with _lib.on_device(y.device):
            _lib.check(_lib.lib().ms_rms_gate_bwd(y.data_ptr(), 0, 1, z.data_ptr(), int(z.dtype == torch.bfloat16), zps, w.data_ptr(),
                                                  eps, dout.data_ptr(), int(dout.dtype == torch.bfloat16), dy.data_ptr(),
                                                  dz.data_ptr(), D, dw.data_ptr(), npix, D,
                                                  _lib.current_stream_ptr(y.device)), "ms_rms_gate_bwd")
        return dy, dz, dw.to(wdtype), None, None


class RMSNormGated(nn.Module):
    """mamba_ssm.ops.triton.layernorm_gated.RMSNorm as constructed at CNN_Mamba.py:430-431:
    norm_before_gate=False -> y = rmsnorm(x * silu(z)) * weight, one group, no bias."""

    def __init__(self, hidden_size, eps=1e-5, group_size=None, norm_before_gate=True, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        self.register_parameter("bias", None)
        self.group_size = group_size
        self.norm_before_gate = norm_before_gate

    def forward(self, x, z=None, out_dtype=None):
        """`out_dtype` (not in the reference signature): the dtype the consumer wants; default x.dtype as the reference."""
        dt = x.dtype if out_dtype is None else out_dtype
        D = x.shape[-1]
        if x.is_cuda and z is not None and not self.norm_before_gate and (self.group_size or D) == D and D <= 1024 \
                and dt in (torch.float32, torch.bfloat16) and tuple(z.shape) == tuple(x.shape):
            return _RMSNormGate.apply(x, z, self.weight, self.eps, dt)
        x = x.float()
        if z is not None and not self.norm_before_gate:
            x = x * F.silu(z.float())
        gs = self.group_size or x.shape[-1]
        xg = x.view(*x.shape[:-1], x.shape[-1] // gs, gs)
        xg = xg * torch.rsqrt(xg.pow(2).mean(-1, keepdim=True) + self.eps)
        out = xg.view_as(x) * self.weight.float()
        if z is not None and self.norm_before_gate:
            out = out * F.silu(z.float())
        return out.to(dt)


class ConvTConvPW(nn.Module):
    """Image-space stem of CNN_Mamba.VSSM (CNN_Mamba.py:43-94): BN -> conv k1 -> BN (same module) -> ReLU -> conv k2 ->
    + identity -> pointwise conv.  (The reference calls torch.flip twice and discards the result, :84,89: a no-op.)"""

    def __init__(self, in_channels, kernel1=3, kernel2=5, kernel3=1, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.in_channels, self.k1, self.k2, self.k3 = in_channels, kernel1, kernel2, kernel3
        self.act = nn.ReLU()
        self.bn = nn.BatchNorm2d(in_channels)
        self.conv1 = nn.Conv2d(in_channels, in_channels, kernel_size=self.k1, stride=1, padding=(self.k1 - 1) // 2)
        self.conv2 = nn.Conv2d(in_channels, in_channels, kernel_size=self.k2, stride=1, padding=(self.k2 - 1) // 2)
        self.PW_conv = nn.Conv2d(in_channels, in_channels, kernel_size=self.k3)

    def _bn(self, x):
        """self.bn(x) in training mode as reductions + one affine pass.  MIOpen's spatial BatchNorm takes 0.5-0.7 ms per call
        on (32,3,224,224) -- 2.4 ms of the SSD variant's 46 ms step for a 3-channel tensor; this is 10x less.  Same outputs
        and the same running-statistics update as nn.BatchNorm2d (momentum form, unbiased running variance)."""
        bn = self.bn
        if not (x.is_cuda and bn.training and bn.track_running_stats and bn.momentum is not None and bn.affine):
            return bn(x)
        xf = x.float()
        var, mean = torch.var_mean(xf, dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():
            n = xf.numel() // xf.shape[1]
            bn.running_mean.mul_(1 - bn.momentum).add_(mean, alpha=bn.momentum)
            bn.running_var.mul_(1 - bn.momentum).add_(var, alpha=bn.momentum * n / max(n - 1, 1))
            bn.num_batches_tracked += 1
        scale = bn.weight.float() * torch.rsqrt(var + bn.eps)
        shift = bn.bias.float() - mean * scale
        return torch.addcmul(shift.view(1, -1, 1, 1), xf, scale.view(1, -1, 1, 1)).to(x.dtype)

    def forward(self, x):
        identity = x
        x = self.conv1(self._bn(x))
        x = self.conv2(self.act(self._bn(x)))
        return self.PW_conv(x + identity)


def _scan_orders(H, W, device):
    """idx[k, l] = pixel visited at step l by direction k (CNN_Mamba.py:494-498 = MedMamba.py:393-395) and its inverse."""
    L = H * W
    l = torch.arange(L, device=device)
    col = (l % H) * W + l // H
    idx = torch.stack([l, col, L - 1 - l, col.flip(0)])
    inv = torch.empty_like(idx)
    inv.scatter_(1, idx, l.expand(4, L))
    return idx, inv


class _SeqFromPixels(torch.autograd.Function):
    """Column ranges of the channel-last conv output -> their four scan sequences (ms_cross_scan_nhwc), one contiguous
    (B, L, 4, width) tensor per range -- exactly the layouts the SSD operator takes (CNN_Mamba.py:506-519) -- and, backward,
    the 4-term sums written straight into the column ranges of ONE gradient tensor (ms_cross_merge_nhwc)."""

    @staticmethod
    def forward(ctx, xc, widths):
        _lib.require_cuda(xc)
        B, H, W, conv = xc.shape
        xc = xc.float().contiguous()
        lib, stream = _lib.lib(), _lib.current_stream_ptr(xc.device)
        outs, col = [], 0
        with _lib.on_device(xc.device):
            for w in widths:
                o = torch.empty((B, H * W, 4, w), device=xc.device, dtype=torch.float32)
                _lib.check(lib.ms_cross_scan_nhwc(xc.data_ptr() + 4 * col, conv, o.data_ptr(), B, H, W, w, stream), "ms_cross_scan_nhwc")
                outs.append(o); col += w
        assert col == conv
        ctx.geom = (B, H, W, conv, tuple(widths))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        B, H, W, conv, widths = ctx.geom
        dev = next(g for g in grads if g is not None).device
        dxc = torch.empty((B, H, W, conv), device=dev, dtype=torch.float32)
        lib, stream = _lib.lib(), _lib.current_stream_ptr(dev)
        col = 0
        with _lib.on_device(dev):
            for w, g in zip(widths, grads):
                if g is None:
                    dxc[..., col:col + w] = 0
                else:
                    g = g.float().contiguous()
                    _lib.check(lib.ms_cross_merge_nhwc(g.data_ptr(), dxc.data_ptr() + 4 * col, conv, B, H, W, w, stream), "ms_cross_merge_nhwc")
                col += w
        return dxc, None


class _PixelsFromSeq(torch.autograd.Function):
    """(B, L, 4, d) per-direction results in scan order -> (B, L, d) in pixel order, ((y0 + y2) + y1) + y3 (CNN_Mamba.py:542-552)."""

    @staticmethod
    def forward(ctx, y, H, W):
        _lib.require_cuda(y)
        B, L, K, d = y.shape
        y = y.float().contiguous()
        out = torch.empty((B, L, d), device=y.device, dtype=torch.float32)
        with _lib.on_device(y.device):
            _lib.check(_lib.lib().ms_cross_merge_nhwc(y.data_ptr(), out.data_ptr(), d, B, H, W, d, _lib.current_stream_ptr(y.device)),
                       "ms_cross_merge_nhwc")
        ctx.geom = (H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        H, W = ctx.geom
        B, L, d = g.shape
        g = g.float().contiguous()
        dy = torch.empty((B, L, 4, d), device=g.device, dtype=torch.float32)
        with _lib.on_device(g.device):
            _lib.check(_lib.lib().ms_cross_scan_nhwc(g.data_ptr(), d, dy.data_ptr(), B, H, W, d, _lib.current_stream_ptr(g.device)),
                       "ms_cross_scan_nhwc")
        return dy, None, None


def ssd_scan_merge(mod, xc):
    """Shared core of SS2D_with_SSD.forward (CNN_Mamba.py:494-552) and CrossMamba.forward.mamba_core
    (CrossMamba_fusion_2b2.py:283-349): xc (B,H,W,conv_dim) = SiLU(dwconv([x | B | C | dt])) channel-last, fp32 ->
    the four directions scanned and merged back to pixel order, (B,H,W,d_ssm) fp32.  `mod` supplies d_ssm, ngroups,
    d_state, nheads, headdim, A_logs, Ds, dt_bias, D_has_hdim, chunk_size."""
    B, H, W, conv_dim = xc.shape
    L, K = H * W, 4
    GN = mod.ngroups * mod.d_state
    # gathered sequences + the chunked evaluation: on the MFMA kernels when they take the shape (headdim 64, one group, 4 N % 64 == 0),
    # else (torch formulation) only for wide states
    kern_ok = SSD_CHUNK_KERNELS and mod.headdim == 64 and mod.ngroups == 1 and (K * mod.d_state) % 64 == 0
    chunked = 0 < SSD_CHUNKED_MIN_STATE <= K * mod.d_state and (kern_ok or K * mod.d_state >= 128)
    if SSD_PIXEL_ORDER and mod.ngroups == 1 and H * W < (1 << 22) and not chunked:
        # native path: the scan kernels take the four pixel orders themselves, one launch per direction's B/C slice
        from .ss2d_fused import ssd_scan_merge_pixel
        return ssd_scan_merge_pixel(xc, -torch.exp(mod.A_logs.float()), mod.Ds, mod.dt_bias.view(-1), mod.d_ssm, mod.d_state,
                                    mod.nheads, mod.headdim, mod.D_has_hdim)
    # sequence path (wide states / ngroups > 1 / MEDSCAN_SSD_PIXEL=0): the four scan orders materialised channel-last
    # (CNN_Mamba.py:494-498), each operand straight in the layout of :506-519 -- heads = (direction, head); B/C = the four
    # directions' states, flat (k, g, n) order regrouped as "(g n)" as :517-519 does
    xs, Bs, Cs, dts = _SeqFromPixels.apply(xc, (mod.d_ssm, GN, GN, mod.nheads))
    As = -torch.exp(mod.A_logs.float())
    Ds = mod.Ds.view(-1, mod.headdim) if mod.D_has_hdim else mod.Ds
    y = mamba_chunk_scan_combined(xs.view(B, L, K * mod.nheads, mod.headdim), dts.view(B, L, K * mod.nheads), As,
                                  Bs.view(B, L, mod.ngroups, -1), Cs.view(B, L, mod.ngroups, -1), chunk_size=mod.chunk_size,
                                  D=Ds, z=None, dt_bias=mod.dt_bias.view(-1), dt_softplus=True)
    assert y.dtype == torch.float
    # cross-merge (CNN_Mamba.py:542-552): every direction back to pixel order and added, ((y1+y2)+y3)+y4
    return _PixelsFromSeq.apply(y.reshape(B, L, K, mod.d_ssm), H, W).view(B, H, W, -1)


def proj(lin, x):
    """`lin(x)` for the bias-free token projections of the SSD blocks (CNN_Mamba.py:441-449,489,564; CrossMamba_fusion_2b2.py:142-160):
    this package's GEMMs (ms_gemm_bf16 under bf16 autocast, ms_gemm_f32 in fp32 -- ss2d_ops.linear_splitk) instead of the library's."""
    # both widths multiples of 8: rows of x, W, dy (and of W^T in the fp32 form) are then whole 16-byte pieces, which is what the kernels
    # read; e.g. in_proj of the first two stages (290 / 548 outputs) stays on the library
    if lin.bias is None and x.is_cuda and lin.weight.shape[0] % 8 == 0 and lin.weight.shape[1] % 8 == 0:
        from .ss2d_ops import linear_splitk
        return linear_splitk(x, lin.weight)
    return lin(x)


def ssd_tail(mod, out, z, z0, x0, d_mlp):
    """RMSNormGated, optional gated-MLP concat, out_proj, dropout (CNN_Mamba.py:554-564)."""
    if mod.rmsnorm:
        # straight into the dtype out_proj consumes when nothing else is concatenated in front of it
        bf16 = d_mlp == 0 and out.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
        out = mod.norm(out, z, out_dtype=torch.bfloat16 if bf16 else None)
    if d_mlp > 0:
        out = torch.cat([F.silu(z0) * x0, out], dim=-1)
    out_data = proj(mod.out_proj, out)
    if mod.dropout is not None:
        out_data = mod.dropout(out_data)
    return out_data


def ssd_dwconv_silu(conv, x, d_conv):
    """SiLU(depthwise conv(x)) on a channel-last (B,H,W,C) tensor (a view is fine) -> (B,H,W,C) fp32."""
    if d_conv == 3:
        return dwconv3x3_silu_nhwc(x, conv.weight, conv.bias)
    return F.silu(conv(x.permute(0, 3, 1, 2))).permute(0, 2, 3, 1).float()


class SS2D_with_SSD(nn.Module):
    """CNN_Mamba.py:322-564.  state_dict: in_proj.weight (2*d_inner + 2*G*N + nheads, d_model), conv2d.{weight
    (d_ssm+2GN+nheads,1,3,3), bias}, dt_bias (4,nheads), A_logs (4*nheads), Ds (4*nheads | 4*d_ssm), norm.weight (d_ssm),
    out_proj.weight (d_model, d_inner)."""

    def __init__(self, d_model, d_state=128, d_conv=3, expand=2, headdim=64, d_ssm=None, ngroups=1,
                 A_init_range=(1, 16), D_has_hdim=False, rmsnorm=True, norm_before_gate=False, dt_rank="auto",
                 dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0, dt_init_floor=1e-4,
                 dt_limit=(0.0, float("inf")), dropout=0., conv_bias=True, bias=False, chunk_size=256,
                 use_mem_eff_path=True, layer_idx=None, process_group=None, sequence_parallel=True, device=None,
                 dtype=None, **kwargs):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        if process_group is not None:
            raise RuntimeError("tensor-parallel SS2D_with_SSD is dead code in the reference (no caller passes "
                               "process_group, CNN_Mamba.py:595) and is not built")
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.process_group, self.sequence_parallel, self.world_size, self.local_rank = None, sequence_parallel, 1, 0
        self.d_inner = int(self.expand * self.d_model)
        self.headdim = headdim
        self.d_ssm = self.d_inner if d_ssm is None else d_ssm
        self.ngroups = ngroups
        assert self.d_ssm % self.headdim == 0
        self.nheads = self.d_ssm // self.headdim
        self.D_has_hdim, self.rmsnorm, self.norm_before_gate = D_has_hdim, rmsnorm, norm_before_gate
        self.dt_limit, self.chunk_size, self.use_mem_eff_path, self.layer_idx = dt_limit, chunk_size, use_mem_eff_path, layer_idx
        self.dt_rank = math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank

        d_in_proj = 2 * self.d_inner + 2 * self.ngroups * self.d_state + self.nheads          # [z, x, B, C, dt]
        self.in_proj = nn.Linear(self.d_model, d_in_proj, bias=bias, **fk)
        conv_dim = self.d_ssm + 2 * self.ngroups * self.d_state + self.nheads
        self.conv2d = nn.Conv2d(conv_dim, conv_dim, groups=conv_dim, bias=conv_bias, kernel_size=d_conv,
                                padding=(d_conv - 1) // 2, **fk)
        self.act = nn.SiLU()
        dt = torch.exp(torch.rand(self.nheads, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = torch.clamp(dt, min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))
        self.dt_bias = nn.Parameter(torch.stack([inv_dt] * 4, dim=0))                          # (4, nheads)
        self.dt_bias._no_weight_decay = True
        self.A_logs = self.A_log_init(A_init_range, self.nheads, dtype, copies=4)              # (4*nheads)
        self.Ds = self.D_init(self.d_ssm, self.D_has_hdim, self.nheads, copies=4)              # (4*nheads) | (4*d_ssm)
        if self.rmsnorm:
            self.norm = RMSNormGated(self.d_ssm, eps=1e-5, norm_before_gate=self.norm_before_gate,
                                     group_size=self.d_ssm // ngroups, **fk)
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **fk)
        self.dropout = nn.Dropout(dropout) if dropout > 0. else None

    @staticmethod
    def A_log_init(A_init_range, nheads, dtype, copies=1, device=None, merge=True):
        assert A_init_range[0] > 0 and A_init_range[1] >= A_init_range[0]
        A_log = torch.log(torch.empty(nheads, dtype=torch.float32, device=device).uniform_(*A_init_range)).to(dtype=dtype)
        if copies > 1:
            A_log = A_log.unsqueeze(0).repeat(copies, 1)
            if merge:
                A_log = A_log.flatten(0, 1)
        A_log = nn.Parameter(A_log)
        A_log._no_weight_decay = True
        return A_log

    @staticmethod
    def D_init(d_ssm, D_has_hdim, nheads, copies=1, device=None, merge=True):
        D = torch.ones(d_ssm if D_has_hdim else nheads, device=device)
        if copies > 1:
            D = D.unsqueeze(0).repeat(copies, 1)
            if merge:
                D = D.flatten(0, 1)
        D = nn.Parameter(D)
        D._no_weight_decay = True
        return D

    def forward(self, u: torch.Tensor, seqlen=None, seq_idx=None, cu_seqlens=None):
        _lib.require_cuda(u)
        B, H, W, C = u.shape
        L, K = H * W, 4
        GN = self.ngroups * self.d_state
        zxbcdt = proj(self.in_proj, u)
        d_mlp = (zxbcdt.shape[-1] - 2 * self.d_ssm - 2 * GN - self.nheads) // 2
        z0, x0, z, xBCdt = torch.split(zxbcdt, [d_mlp, d_mlp, self.d_ssm, self.d_ssm + 2 * GN + self.nheads], dim=-1)
        # depthwise conv + SiLU over the whole [x | B | C | dt] stack (dt goes through the conv too, CNN_Mamba.py:490-491),
        # channel-last, reading zxbcdt in place
        xc = ssd_dwconv_silu(self.conv2d, xBCdt, self.d_conv)                                   # (B,H,W,conv_dim) fp32
        return ssd_tail(self, ssd_scan_merge(self, xc), z, z0, x0, d_mlp)


class SS_Conv_SSD(nn.Module):
    """CNN_Mamba.py:583-619: same two-branch block as SS_Conv_SSM with the SSD mixer on the right half."""

    def __init__(self, hidden_dim: int = 0, drop_path: float = 0,
                 norm_layer: Callable[..., torch.nn.Module] = partial(nn.LayerNorm, eps=1e-6),
                 attn_drop_rate: float = 0, d_state: int = 64, **kwargs):
        super().__init__()
        half = hidden_dim // 2
        self.ln_1 = norm_layer(half)
        self.self_attention = SS2D_with_SSD(d_model=half, dropout=attn_drop_rate, d_state=d_state, **kwargs)
        self.drop_path = DropPath(drop_path)
        self.conv33conv33conv11 = nn.Sequential(
            nn.BatchNorm2d(half), nn.Conv2d(half, half, kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(half), nn.ReLU(), nn.Conv2d(half, half, kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(half), nn.ReLU(), nn.Conv2d(half, half, kernel_size=1, stride=1), nn.ReLU())

    def forward(self, input: torch.Tensor):
        if mm.BLOCK_FUSED and input.is_cuda and input.shape[-1] % 4 == 0 and type(self.ln_1) is nn.LayerNorm \
                and self.ln_1.elementwise_affine and self.ln_1.bias is not None:
            return mm.fused_block_forward(self, input)
        left, right = input.chunk(2, dim=-1)
        x = self.drop_path(self.self_attention(self.ln_1(right)))
        if CONV_CHANNELS_LAST and left.is_cuda:
            left = self.conv33conv33conv11(left.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last))
            left = left.permute(0, 2, 3, 1)
        else:
            left = self.conv33conv33conv11(left.permute(0, 3, 1, 2).contiguous()).permute(0, 2, 3, 1).contiguous()
        return channel_shuffle(torch.cat((left, x), dim=-1), groups=2) + input


class VSSLayer(nn.Module):
    """CNN_Mamba.py:622-685."""

    def __init__(self, dim, depth, attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None,
                 use_checkpoint=False, d_state=64, **kwargs):
        super().__init__()
        self.dim, self.use_checkpoint = dim, use_checkpoint
        self.blocks = nn.ModuleList([
            SS_Conv_SSD(hidden_dim=dim, drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                        norm_layer=norm_layer, attn_drop_rate=attn_drop, d_state=d_state) for i in range(depth)])
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x):
        for blk in self.blocks:
            x = checkpoint.checkpoint(blk, x, use_reentrant=False) if self.use_checkpoint else blk(x)
        return x if self.downsample is None else self.downsample(x)


class VSSM(nn.Module):
    """CNN_Mamba.py:752-851: defaults dims [128,256,512,1024], d_state 16 (-> SSD state 4*16 = 64), ConvTConvPW stem."""

    def __init__(self, patch_size=4, in_chans=3, num_classes=1000, depths=[2, 2, 4, 2], depths_decoder=[2, 9, 2, 2],
                 dims=[128, 256, 512, 1024], dims_decoder=[1024, 512, 256, 128], d_state=16, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, patch_norm=True,
                 use_checkpoint=False, **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.num_layers = len(depths)
        if isinstance(dims, int):
            dims = [int(dims * 2 ** i) for i in range(self.num_layers)]
        self.embed_dim, self.num_features, self.dims = dims[0], dims[-1], dims
        self.patch_embed = PatchEmbed2D(patch_size=patch_size, in_chans=in_chans, embed_dim=self.embed_dim,
                                        norm_layer=norm_layer if patch_norm else None)
        self.ape = False
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [r.item() for r in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(VSSLayer(
                dim=dims[i], depth=depths[i], d_state=math.ceil(dims[0] / 6) if d_state is None else d_state,
                drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                norm_layer=norm_layer, downsample=PatchMerging2D if (i < self.num_layers - 1) else None,
                use_checkpoint=use_checkpoint))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.conv_T_conv = ConvTConvPW(in_channels=in_chans)
        self.apply(self._init_weights)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _init_weights(self, m: nn.Module):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"absolute_pos_embed"}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {"relative_position_bias_table"}

    def forward_backbone(self, x):
        x = self.pos_drop(self.patch_embed(x))
        for layer in self.layers:
            x = layer(x)
        return x

    def forward(self, x):
        x = self.forward_backbone(self.conv_T_conv(x))
        x = self.avgpool(x.permute(0, 3, 1, 2))
        return self.head(torch.flatten(x, start_dim=1))


MedSSD = SS2D_with_SSD      # the name CrossMamba/CrossMamba_fusion_2b2.py:390 gives the same class
