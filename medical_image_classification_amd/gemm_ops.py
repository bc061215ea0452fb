"""The dense projections of SS2D (in_proj / x_proj / out_proj, MedMamba.py:284,326,397,469,480) on the hand-written
MFMA kernel `ms_gemm_bf16` (csrc/gemm.hip; C ABI in include/medscan.h): bf16 matrix cores, fp32 accumulation.

  forward      y  = x @ W^T        activations bf16 or fp32 in memory (fp32 is rounded to bf16 while a tile is staged),
  input grad   dx = dy @ W         the weight is read as the fp32 master copy -- no bf16 weight copies, no cast kernels
  weight grad  dW = dy^T @ x       reduction over the B*H*W tokens split over workgroups inside the kernel (fp32 atomics)

This is the autocast-bf16 path (what the reference's nn.Linear / einsum do under `torch.autocast(bfloat16)`); fp32 runs keep
the stock fp32 GEMMs.  CUDA (HIP) tensors only -- no CPU fallback.
"""
import os

import torch

from . import _lib, arena, shadow


def _is_f32(t):
    if t.dtype == torch.float32:
        return 1
    if t.dtype == torch.bfloat16:
        return 0
    raise RuntimeError(f"ms_gemm_bf16: bf16 or fp32 operands only, got {t.dtype}")


def _k_splits(k_len, out_blocks, out_elems):
    """Slices of the reduction for the weight gradient: about one workgroup per CU in total, every slice at least 256 deep,
    and at most ~6 MB of fp32 atomics over all slices (measured on MI355X, tools/sweep_gemm_dw.py: more slices lose to
    same-address atomic contention and atomic volume, fewer to exposed load latency)."""
    want = max(1, 256 // max(1, out_blocks))
    return max(1, min(want, k_len // 256, max(1, 1_500_000 // max(1, out_elems))))


def _blocks(rows, cols):
    """Coarse tile count of a (rows x cols) weight-gradient output (128-row blocks, 64/128/192-column blocks) that the split
    heuristic below was tuned with; the kernel itself runs 64 x 64 tiles, i.e. ~4x as many workgroups."""
    passes = (cols + 191) // 192
    per = -(-cols // passes)
    bn = 64 if per <= 64 else 128 if per <= 128 else 192
    return -(-rows // 128) * -(-cols // bn)


def wgrad_bias_ok(dy, x):
    return dy.shape[1] >= x.shape[1]


def weight_grad(dy, x, out=None, dbias=None):
    """dW (N, K) (+)= dy^T @ x for dy (M, N), x (M, K): split-K inside the kernel, the taller of (N, K) on the row side.
    dbias (N,) fp32, zero-initialised: also receives dy's column sums from the same launch (ms_gemm_bf16_wgrad_bias; only in the
    orientation with dy on the row side, N >= K -- the caller checks `wgrad_bias_ok`)."""
    N, K = dy.shape[1], x.shape[1]
    M = dy.shape[0]
    if out is None:
        out = arena.zeros((N, K), dy.device)
    if dbias is not None:
        if not wgrad_bias_ok(dy, x):
            raise RuntimeError("ms_gemm_bf16_wgrad_bias: needs N >= K")
        _lib.require_cuda(dy, x)
        with _lib.on_device(dy.device):
            _lib.check(_lib.lib().ms_gemm_bf16_wgrad_bias(dy.data_ptr(), _is_f32(dy), dy.stride(0), x.data_ptr(), _is_f32(x), x.stride(0),
                                                          out.data_ptr(), out.stride(0), dbias.data_ptr(), N, K, M,
                                                          _k_splits(M, _blocks(N, K), N * K), _lib.current_stream_ptr(dy.device)),
                       "ms_gemm_bf16_wgrad_bias")
        return out
    if N >= K:
        return gemm(dy, x, a_trans=True, b_trans=True, out=out, accumulate=True, k_splits=_k_splits(M, _blocks(N, K), N * K))
    # (K x N) orientation, accumulated into the transposed output
    _lib.require_cuda(dy, x)
    ks = _k_splits(M, _blocks(K, N), N * K)
    with _lib.on_device(dy.device):
        _lib.check(_lib.lib().ms_gemm_bf16(x.data_ptr(), _is_f32(x), 1, x.stride(0), dy.data_ptr(), _is_f32(dy), 1, dy.stride(0),
                                           out.data_ptr(), 3, out.stride(0), K, N, M, ks, _lib.current_stream_ptr(dy.device)),
                   "ms_gemm_bf16")
    return out


def gemm(a, b, a_trans=False, b_trans=False, out=None, out_dtype=torch.float32, accumulate=False, k_splits=1, bias=None, relu=False):
    """C[i, j] (+)= sum_k Aop[i, k] * Bop[j, k];  a, b: 2-D row-major (unit inner stride) bf16 / fp32 CUDA tensors,
    Aop = a.T if a_trans else a, Bop = b.T if b_trans else b.  Returns C (M, N) in `out_dtype` (fp32 when accumulating).
    bias (N,) fp32 / relu: epilogue C = [relu](.. + bias[j]) of the store modes (ms_gemm_bf16_bias_act)."""
    _lib.require_cuda(a, b)
    if a.dim() != 2 or b.dim() != 2 or a.stride(1) != 1 or b.stride(1) != 1:
        raise RuntimeError("ms_gemm_bf16: 2-D operands with unit inner stride")
    M, K = (a.shape[1], a.shape[0]) if a_trans else a.shape
    N, Kb = (b.shape[1], b.shape[0]) if b_trans else b.shape
    if K != Kb:
        raise RuntimeError(f"ms_gemm_bf16: inner dimensions differ ({K} vs {Kb})")
    if accumulate or k_splits > 1:
        c_mode = 2
        if out is None:
            out = arena.zeros((M, N), a.device)
    else:
        c_mode = 1 if out_dtype == torch.bfloat16 else 0
        if out is None:
            out = torch.empty((M, N), device=a.device, dtype=torch.bfloat16 if c_mode == 1 else torch.float32)
    if out.stride(1) != 1 or tuple(out.shape) != (M, N):
        raise RuntimeError("ms_gemm_bf16: bad output tensor")
    with _lib.on_device(a.device):
        if bias is not None or relu:
            if c_mode == 2:
                raise RuntimeError("ms_gemm_bf16_bias_act: the epilogue belongs to the store modes (no accumulation / split-K)")
            if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N or not bias.is_contiguous() or not bias.is_cuda):
                raise RuntimeError("ms_gemm_bf16_bias_act: bias must be a contiguous fp32 CUDA vector of N entries")
            _lib.check(_lib.lib().ms_gemm_bf16_bias_act(a.data_ptr(), _is_f32(a), int(a_trans), a.stride(0), b.data_ptr(), _is_f32(b),
                                                        int(b_trans), b.stride(0), out.data_ptr(), c_mode, out.stride(0), M, N, K,
                                                        bias.data_ptr() if bias is not None else None, int(bool(relu)),
                                                        _lib.current_stream_ptr(a.device)), "ms_gemm_bf16_bias_act")
        else:
            _lib.check(_lib.lib().ms_gemm_bf16(a.data_ptr(), _is_f32(a), int(a_trans), a.stride(0), b.data_ptr(), _is_f32(b), int(b_trans),
                                               b.stride(0), out.data_ptr(), c_mode, out.stride(0), M, N, K, int(k_splits),
                                               _lib.current_stream_ptr(a.device)), "ms_gemm_bf16")
    return out


_FUSED_BWD = os.environ.get("MEDSCAN_LINEAR_BWD_FUSED", "1") == "1"
_FUSED_BWD_MIN_ROWS = int(os.environ.get("MEDSCAN_LINEAR_BWD_MIN_ROWS", "16384"))
_FUSED_BWD_MIN_ROW_BYTES = int(os.environ.get("MEDSCAN_LINEAR_BWD_MIN_ROW_BYTES", "512"))


def linear_bwd_fused_ok(dy, x, w):
    """Both backward products of y = x W^T from one pass (ms_linear_bwd_bf16): 2-D operands the kernel reads in place, a width pair it is
    built for, and enough token rows that streaming dy / x once is what matters (the early stages)."""
    if not (_FUSED_BWD and dy.is_cuda and dy.dim() == 2 and x.dim() == 2 and w.dim() == 2 and w.is_contiguous() and dy.shape[0] >= _FUSED_BWD_MIN_ROWS):
        return False
    N, K = w.shape
    if dy.shape[1] != N or x.shape[1] != K or x.shape[0] != dy.shape[0]:
        return False
    for t in (dy, x):
        g = 4 if t.dtype == torch.float32 else 8
        if t.dtype not in (torch.float32, torch.bfloat16) or t.stride(1) != 1 or t.stride(0) % g != 0 or t.data_ptr() % 16 != 0:
            return False
    if w.dtype not in (torch.float32, torch.bfloat16) or w.data_ptr() % 16 != 0:
        return False
    if N % (4 if dy.dtype == torch.float32 else 8) != 0:
        return False
    # rows of at least 512 bytes of dy + x: below that (in_proj / out_proj of MedMamba-T's stage 0: 480 / 384 B) the two-launch form, whose
    # operands then mostly come from the caches, measured faster in situ (50 / 41 us against 57 / 46); x_proj (944 B): 166 -> 101 us
    if N * dy.element_size() + K * x.element_size() < _FUSED_BWD_MIN_ROW_BYTES:
        return False
    return bool(_lib.lib().ms_linear_bwd_ok(N, K))


def linear_bwd_fused(dy, x, w, dx_dtype=torch.float32):
    """(dx, dW) = (dy @ W, dy^T @ x): dx (M, K) in `dx_dtype` (fp32 / bf16), dW (N, K) fp32.  See linear_bwd_fused_ok."""
    M, N = dy.shape
    K = x.shape[1]
    dx = torch.empty((M, K), device=dy.device, dtype=dx_dtype)
    dw = arena.zeros((N, K), dy.device)
    with _lib.on_device(dy.device):
        _lib.check(_lib.lib().ms_linear_bwd_bf16(dy.data_ptr(), _is_f32(dy), dy.stride(0), x.data_ptr(), _is_f32(x), x.stride(0), w.data_ptr(),
                                                 _is_f32(w), dx.data_ptr(), int(dx_dtype == torch.bfloat16), dx.stride(0), dw.data_ptr(), M, N, K,
                                                 _lib.current_stream_ptr(dy.device)), "ms_linear_bwd_bf16")
    return dx, dw


def _rows(t):
    """(.., K) -> (M, K) view usable by the kernel (unit inner stride, uniform row stride) or a contiguous copy."""
    t2 = t.reshape(-1, t.shape[-1])
    if t2.dtype not in (torch.float32, torch.bfloat16):
        t2 = t2.float()
    if t2.dtype == torch.bfloat16 and t2.shape[1] % 8 != 0:
        t2 = t2.float()                      # rows must be multiples of 16 bytes: such widths go through as fp32
    esz = 4 if t2.dtype == torch.float32 else 8
    if t2.stride(1) != 1 or t2.stride(0) % esz != 0 or t2.data_ptr() % 16 != 0:
        t2 = t2.contiguous()
    return t2


class _LinearMFMA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, out_fp32):
        xm = _rows(x)
        if weight.dtype == torch.float32 and weight.is_contiguous() and weight.shape[1] % 8 == 0:
            w = shadow.bf16(weight)          # cached bf16 copy (one refresh launch per step for all weights): half the cold bytes
        else:
            w = weight.detach().float().contiguous()
        y = gemm(xm, w, out_dtype=torch.float32 if out_fp32 else torch.bfloat16)
        ctx.save_for_backward(xm, w)
        ctx.xshape, ctx.xdtype, ctx.wdtype = x.shape, x.dtype, weight.dtype
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xm, w = ctx.saved_tensors
        shadow.invalidate(xm.device)             # a backward pass is under way: the weights are about to change
        N, K = w.shape
        dym = _rows(dy)
        dx = None
        if ctx.needs_input_grad[0] and linear_bwd_fused_ok(dym, xm, w):
            dx, dw = linear_bwd_fused(dym, xm, w, torch.bfloat16 if ctx.xdtype == torch.bfloat16 else torch.float32)
            return dx.view(ctx.xshape), dw.to(ctx.wdtype), None
        if ctx.needs_input_grad[0]:
            dx = gemm(dym, w, b_trans=True, out_dtype=torch.bfloat16 if ctx.xdtype == torch.bfloat16 else torch.float32)
            dx = dx.view(ctx.xshape)
        dw = weight_grad(dym, xm)
        return dx, dw.to(ctx.wdtype), None


def linear_mfma(x, weight, out_fp32=False):
    """F.linear(x, weight) (no bias) on ms_gemm_bf16: bf16 MFMA with fp32 accumulation; bf16 result unless out_fp32."""
    return _LinearMFMA.apply(x, weight, out_fp32)


# ---- fp32: the reference's own precision (it never uses autocast, train.py:57-77) on the exact-fp32 matrix instruction ----------
def gemm_f32(a, b, a_trans=False, b_trans=False, out=None, accumulate=False, k_splits=1, bias=None, relu=False):
    """C[i, j] (+)= sum_k Aop[i, k] * Bop[j, k] with fp32 operands, products and sums (ms_gemm_f32: v_mfma_f32_16x16x4_f32); a, b 2-D
    row-major fp32 CUDA tensors with unit inner stride.  accumulate / k_splits > 1: fp32 atomics into a zero-initialised (or the given) C."""
    _lib.require_cuda(a, b)
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.dim() != 2 or b.dim() != 2 or a.stride(1) != 1 or b.stride(1) != 1:
        raise RuntimeError("ms_gemm_f32: 2-D fp32 operands with unit inner stride")
    M, K = (a.shape[1], a.shape[0]) if a_trans else a.shape
    N, Kb = (b.shape[1], b.shape[0]) if b_trans else b.shape
    if K != Kb:
        raise RuntimeError(f"ms_gemm_f32: inner dimensions differ ({K} vs {Kb})")
    c_mode = 2 if (accumulate or k_splits > 1) else 0
    if out is None:
        out = arena.zeros((M, N), a.device) if c_mode == 2 else torch.empty((M, N), device=a.device, dtype=torch.float32)
    if out.stride(1) != 1 or tuple(out.shape) != (M, N) or out.dtype != torch.float32:
        raise RuntimeError("ms_gemm_f32: bad output tensor")
    with _lib.on_device(a.device):
        _lib.check(_lib.lib().ms_gemm_f32(a.data_ptr(), int(a_trans), a.stride(0), b.data_ptr(), int(b_trans), b.stride(0), out.data_ptr(),
                                          c_mode, out.stride(0), M, N, K, int(k_splits), bias.data_ptr() if bias is not None else None,
                                          int(bool(relu)), _lib.current_stream_ptr(a.device)), "ms_gemm_f32")
    return out


def weight_grad_f32(dy, x, out=None):
    """dW (N, K) (+)= dy^T @ x in fp32: split-K inside the kernel, the taller of (N, K) on the row side."""
    N, K, M = dy.shape[1], x.shape[1], dy.shape[0]
    if out is None:
        out = arena.zeros((N, K), dy.device)
    if N >= K:
        return gemm_f32(dy, x, a_trans=True, b_trans=True, out=out, accumulate=True, k_splits=_k_splits(M, _blocks(N, K), N * K))
    _lib.require_cuda(dy, x)
    with _lib.on_device(dy.device):
        _lib.check(_lib.lib().ms_gemm_f32(x.data_ptr(), 1, x.stride(0), dy.data_ptr(), 1, dy.stride(0), out.data_ptr(), 3, out.stride(0),
                                          K, N, M, _k_splits(M, _blocks(K, N), N * K), None, 0, _lib.current_stream_ptr(dy.device)),
                   "ms_gemm_f32")
    return out


def f32_rows_ok(t):
    """rows the fp32 kernel can read in place: fp32, unit inner stride, row stride a multiple of 4 floats, 16-byte aligned."""
    return t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def _rows_f32(t):
    t2 = t.reshape(-1, t.shape[-1])
    if t2.dtype != torch.float32:
        t2 = t2.float()
    return t2 if f32_rows_ok(t2) else t2.contiguous()


class _LinearF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        xm = _rows_f32(x)
        w = weight.detach()
        w = w if f32_rows_ok(w) else w.float().contiguous()
        y = gemm_f32(xm, w)
        ctx.save_for_backward(xm, w)
        ctx.xshape, ctx.xdtype, ctx.wdtype = x.shape, x.dtype, weight.dtype
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xm, w = ctx.saved_tensors
        dym = _rows_f32(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            # dx = dy @ W as a plain / plain product against W^T (one small transposing copy of the weight): a transposed operand's
            # fragments are four ds_read_b32 instead of one ds_read_b128, which made this product LDS-instruction bound (in_proj at
            # stage 1: 137 -> see tools/bench_gemm_f32.py)
            dx = gemm_f32(dym, w.t().contiguous()).view(ctx.xshape).to(ctx.xdtype)
        return dx, weight_grad_f32(dym, xm).to(ctx.wdtype)


def linear_f32(x, weight):
    """F.linear(x, weight) (no bias) in fp32 on ms_gemm_f32 -- forward, input gradient and split-K weight gradient."""
    return _LinearF32.apply(x, weight)
