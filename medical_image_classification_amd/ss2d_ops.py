"""torch.autograd wrappers of the non-scan HIP kernels on the SS2D path (C ABI: include/medscan.h).

Each wrapper replaces a run of eager tensor ops of the reference's `SS2D`:
  dwconv3x3_silu  <- `self.act(self.conv2d(x))`                         (MedMamba.py:285-294,473)
  cross_scan      <- stack / transpose / flip / cat                      (MedMamba.py:393-395)
  cross_merge     <- flips / transposes and `y1 + y2 + y3 + y4`          (MedMamba.py:420-424,476)
cross_scan and cross_merge are each other's adjoint, so each one's backward is the other's forward.
All of them need CUDA (HIP) tensors and raise RuntimeError otherwise -- no CPU fallback.
"""
import torch

from . import _lib


def _stream(t):
    return _lib.current_stream_ptr(t.device)


class _CrossScan(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x):
        _lib.require_cuda(x)
        B, D, H, W = x.shape
        x = x.contiguous()
        xs = torch.empty((B, 4, D, H * W), device=x.device, dtype=torch.float32)
        ctx.hw = (H, W)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ms_cross_scan(x.data_ptr(), xs.data_ptr(), B, D, H, W, _stream(x)), "ms_cross_scan")
        return xs

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        H, W = ctx.hw
        B, _, D, L = g.shape
        g = g.contiguous().float()
        dx = torch.empty((B, D, H, W), device=g.device, dtype=torch.float32)
        with _lib.on_device(g.device):
            _lib.check(_lib.lib().ms_cross_merge(g.data_ptr(), dx.data_ptr(), B, D, H, W, _stream(g)), "ms_cross_merge")
        return dx


class _CrossMerge(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, ys, H, W):
        _lib.require_cuda(ys)
        B, K, D, L = ys.shape
        if K != 4 or L != H * W:
            raise RuntimeError(f"cross_merge: expected (B,4,D,{H * W}), got {tuple(ys.shape)}")
        ys = ys.contiguous()
        y = torch.empty((B, D, L), device=ys.device, dtype=torch.float32)
        ctx.hw = (H, W)
        with _lib.on_device(ys.device):
            _lib.check(_lib.lib().ms_cross_merge(ys.data_ptr(), y.data_ptr(), B, D, H, W, _stream(ys)), "ms_cross_merge")
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        H, W = ctx.hw
        B, D, L = g.shape
        g = g.contiguous().float()
        dys = torch.empty((B, 4, D, L), device=g.device, dtype=torch.float32)
        with _lib.on_device(g.device):
            _lib.check(_lib.lib().ms_cross_scan(g.data_ptr(), dys.data_ptr(), B, D, H, W, _stream(g)), "ms_cross_scan")
        return dys, None, None


class _DWConvSiLU(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias):
        _lib.require_cuda(x, weight, bias)
        B, C, H, W = x.shape
        if tuple(weight.shape) != (C, 1, 3, 3):
            raise RuntimeError(f"dwconv3x3_silu: weight must be ({C},1,3,3), got {tuple(weight.shape)}")
        x, weight = x.contiguous(), weight.contiguous()
        bias = bias.contiguous() if bias is not None else None
        y = torch.empty_like(x)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ms_dwconv3x3_silu_fwd(x.data_ptr(), weight.data_ptr(),
                                                       bias.data_ptr() if bias is not None else None,
                                                       y.data_ptr(), B, C, H, W, _stream(x)), "ms_dwconv3x3_silu_fwd")
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        B, C, H, W = x.shape
        dy = dy.contiguous().float()
        dx = torch.empty_like(x)
        dw = torch.zeros_like(weight)
        db = torch.zeros_like(bias) if bias is not None else None
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ms_dwconv3x3_silu_bwd(x.data_ptr(), weight.data_ptr(),
                                                       bias.data_ptr() if bias is not None else None,
                                                       dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                                       db.data_ptr() if db is not None else None,
                                                       B, C, H, W, _stream(x)), "ms_dwconv3x3_silu_bwd")
        return dx, dw, db


def cross_scan(x):
    """(B, D, H, W) -> (B, 4, D, H*W): row-major, column-major and both reversed (MedMamba.py:393-395)."""
    return _CrossScan.apply(x)


def cross_merge(ys, H, W):
    """(B, 4, D, H*W) -> (B, D, H*W) = ((y0 + flip(y2)) + T(y1)) + T(flip(y3)) (MedMamba.py:420-424,476)."""
    return _CrossMerge.apply(ys, H, W)


def dwconv3x3_silu(x, weight, bias):
    """SiLU(depthwise conv3x3(x) + bias), NCHW (MedMamba.py:285-294,473)."""
    return _DWConvSiLU.apply(x, weight, bias)


class _LinearSplitK(torch.autograd.Function):
    """y = x @ W^T for token matrices with a huge row count M = B*H*W and small feature dims (the in/out/x projections
    of SS2D: M = 200 704, K/N = 48..192 at stage 0).  Forward and dX are ordinary GEMMs; the weight gradient
    dW = dY^T @ X has an (N x K) output of a few tiles and a reduction over M, which the BLAS heuristics run on 3-12
    workgroups.  Here the reduction is split into S slices evaluated as one batched GEMM (S*tiles workgroups) and summed
    in fp32 -- split-K by construction."""

    @staticmethod
    def forward(ctx, x, weight, out_fp32=False):
        ac = torch.is_autocast_enabled()
        dt = torch.get_autocast_dtype("cuda") if ac else None
        xm = x.reshape(-1, x.shape[-1])
        if ac:
            from . import shadow
            xm = xm.to(dt)
            w = shadow.bf16(weight) if (dt == torch.bfloat16 and weight.dtype == torch.float32) else weight.detach().to(dt)
        else:
            w = weight
        with torch.autocast(device_type="cuda", enabled=False):
            # out_fp32: the consumer wants fp32 (the scan operands): let the GEMM write it instead of casting afterwards
            y = torch.mm(xm, w.t(), out_dtype=torch.float32) if (out_fp32 and xm.dtype != torch.float32) else torch.mm(xm, w.t())
        ctx.save_for_backward(xm, w)
        ctx.xshape, ctx.wdtype, ctx.xdtype = x.shape, weight.dtype, x.dtype
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xm, w = ctx.saved_tensors
        from . import shadow
        shadow.invalidate(xm.device)             # a backward pass is under way: the weights are about to change
        M, K = xm.shape
        N = w.shape[0]
        dym = dy.reshape(M, N).to(xm.dtype)
        with torch.autocast(device_type="cuda", enabled=False):
            dx = torch.mm(dym, w) if ctx.needs_input_grad[0] else None
            S = 1
            for cand in (128, 64, 32, 16, 8, 4, 2):
                if M % cand == 0 and M // cand >= 1024:
                    S = cand
                    break
            if S > 1:
                a, b = dym.view(S, M // S, N).transpose(1, 2), xm.view(S, M // S, K)
                part = torch.bmm(a, b, out_dtype=torch.float32) if xm.dtype != torch.float32 else torch.bmm(a, b)   # (S, N, K)
                dw = part.sum(dim=0)
            else:
                dw = torch.mm(dym.t(), xm).float()
        return (dx.view(ctx.xshape).to(ctx.xdtype) if dx is not None else None), dw.to(ctx.wdtype), None


import os

# the bf16-autocast projections run on the hand-written MFMA kernel (gemm_ops / csrc/gemm.hip); MEDSCAN_MFMA_GEMM=0 puts them
# back on the BLAS library (kernel-variant experiments)
_MFMA_GEMM = os.environ.get("MEDSCAN_MFMA_GEMM", "1") == "1"
# fp32 runs: the projections on ms_gemm_f32 (v_mfma_f32_16x16x4_f32); MEDSCAN_F32_GEMM=0 keeps the library GEMMs (A/B runs)
_F32_GEMM = os.environ.get("MEDSCAN_F32_GEMM", "1") == "1"
# token-matrix height from which the MFMA kernel is used (below it the library GEMM wins in situ: measured)
_MFMA_MIN_ROWS = int(os.environ.get("MEDSCAN_MFMA_MIN_ROWS", "0"))


def linear_splitk(x, weight, out_fp32=False):
    """F.linear(x, weight) (no bias) for the token matrices of SS2D (in_proj / out_proj / PatchMerging reduction,
    MedMamba.py:284,326,208).  Under bf16 autocast: ms_gemm_bf16 (hand-written MFMA kernel: fp32 master weight read
    directly, split-K weight gradient inside the kernel).  fp32: library GEMMs with a split-K weight gradient; F.linear for
    small row counts.  out_fp32: return fp32 even when the GEMM runs in bf16 under autocast."""
    if (_MFMA_GEMM and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and weight.shape[1] % 8 == 0 and x.numel() // x.shape[-1] >= _MFMA_MIN_ROWS):
        from .gemm_ops import linear_mfma
        with torch.autocast(device_type="cuda", enabled=False):
            return linear_mfma(x, weight, out_fp32)
    if (_MFMA_GEMM and _F32_GEMM and x.is_cuda and not torch.is_autocast_enabled() and x.dtype == torch.float32
            and weight.dtype == torch.float32 and weight.shape[1] % 4 == 0 and weight.shape[0] % 4 == 0):     # rows of x, dy and W^T: multiples of 16 bytes
        from .gemm_ops import linear_f32           # fp32 runs (the reference's precision): exact-fp32 MFMA kernel, no library GEMM
        return linear_f32(x, weight)
    if x.is_cuda and (x.numel() // x.shape[-1] >= 8192 or (torch.is_autocast_enabled() and x.requires_grad)):
        return _LinearSplitK.apply(x, weight, out_fp32)
    y = torch.nn.functional.linear(x, weight)
    return y.float() if out_fp32 else y
