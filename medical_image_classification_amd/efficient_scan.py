"""Stride-2 four-way scan of FusionMamba (CrossMamba/FusionMamba/models/cross.py:30-414): the second consumer
of the S6 operator in the reference.  Same names and call signatures as cross.py:

    EfficientScan.apply(x (B,C,H,W), step_size=2)        -> xs (B,4,C,ceil(H/2)*ceil(W/2))      cross.py:139-190
    EfficientMerge.apply(ys (B,4,C,L'), ori_h, ori_w, 2) -> y (B,C,H*W)                         cross.py:30-88
    SelectiveScan.apply(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)               cross.py:91-137
    cross_selective_scan[_new](x, x_proj_weight, x_proj_bias, dt_projs_weight, dt_projs_bias, A_logs, Ds, out_norm, ...)
    cross_selective_scan_cross(x1, x2, ...)                                                    cross.py:193-414

The image is split into its four (row parity, column parity) sub-lattices, zero-padded to even size; each is one scan
sequence of a quarter of the pixels:
    k = 0: rows 2i,   cols 2j,   row-major          k = 1: rows 2i+1, cols 2j,   column-major
    k = 2: rows 2i,   cols 2j+1, row-major          k = 3: rows 2i+1, cols 2j+1, column-major
`EfficientScan` / `EfficientMerge` themselves: both directions of the mapping are ONE gather each (every pixel belongs to exactly
one (k, l); padded positions read as zero): bit-exact, no strided slice assignments.

`cross_selective_scan*` (what the modules call) never materialise the sequences on CUDA: the scan kernels take the four
sub-lattices as an ADDRESSING MODE (`MS_SCAN_LATTICE`, include/medscan.h; `PosMap` modes 4-7 in csrc/scan_common.h) -- the
activations are read in place from the channel-last image, x_proj / dt_proj run per pixel (no gathers in front of them), and
every pixel of the result is written by exactly one group, so the merge costs nothing.  Maps with an odd size are zero-padded to
even sizes first (the reference pads the sequences, cross.py:148-156: the same zeros).  `nrows` is a tiling detail of the
reference's CUDA kernel and is ignored.  MEDSCAN_LATTICE_KERNEL=0 (or CPU tensors / d_state > 16) takes the gather formulation.
"""
import math
import os

import torch
import torch.nn.functional as F

from .selective_scan_interface import selective_scan_fn

_LATTICE_KERNEL = os.environ.get("MEDSCAN_LATTICE_KERNEL", "1") == "1"

_IDX_CACHE = {}


def _lattice_index(H, W, step, device):
    """(to_seq, to_img): to_seq[k*L2 + l] = flat pixel h*W+w visited at step l of sub-lattice k, or H*W for a padded position
    (reads a zero appended to the image); to_img[h*W+w] = k*L2 + l of the pixel."""
    key = (H, W, step, str(device))
    if key not in _IDX_CACHE:
        if step != 2:
            raise RuntimeError("EfficientScan/EfficientMerge: the reference's four sub-lattices exist for step_size 2 only")
        H2, W2 = math.ceil(H / 2), math.ceil(W / 2)
        i, j = torch.meshgrid(torch.arange(H2), torch.arange(W2), indexing="ij")           # row-major (i, j)
        a, b = torch.meshgrid(torch.arange(W2), torch.arange(H2), indexing="ij")           # column-major: a over cols, b over rows
        hw = [(2 * i, 2 * j), (2 * b + 1, 2 * a), (2 * i, 2 * j + 1), (2 * b + 1, 2 * a + 1)]
        to_seq = torch.stack([torch.where((h < H) & (w < W), h * W + w, torch.full_like(h, H * W)).reshape(-1) for h, w in hw]).reshape(-1)
        to_img = torch.empty(H * W + 1, dtype=torch.long)
        to_img[to_seq] = torch.arange(to_seq.numel())
        _IDX_CACHE[key] = (to_seq.to(device), to_img[:H * W].contiguous().to(device), H2 * W2)
    return _IDX_CACHE[key]


def _image_to_sequences(x, step):
    B, C, H, W = x.shape
    to_seq, _, L2 = _lattice_index(H, W, step, x.device)
    flat = torch.cat([x.reshape(B, C, H * W), x.new_zeros(B, C, 1)], dim=-1)
    return flat[:, :, to_seq].view(B, C, 4, L2).transpose(1, 2).contiguous()               # (B,4,C,L2)


def _sequences_to_image(ys, H, W, step):
    B, K, C, L2 = ys.shape
    _, to_img, _ = _lattice_index(H, W, step, ys.device)
    return ys.transpose(1, 2).reshape(B, C, K * L2)[:, :, to_img]                          # (B,C,H*W)


class EfficientScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, step_size=2):
        ctx.shape, ctx.step_size = x.shape, step_size
        return _image_to_sequences(x, step_size)

    @staticmethod
    def backward(ctx, grad_xs):
        B, C, H, W = ctx.shape
        return _sequences_to_image(grad_xs.reshape(B, 4, C, -1), H, W, ctx.step_size).view(B, C, H, W), None


class EfficientMerge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ys, ori_h, ori_w, step_size=2):
        ctx.geom = (int(ori_h), int(ori_w), step_size)
        return _sequences_to_image(ys, int(ori_h), int(ori_w), step_size)

    @staticmethod
    def backward(ctx, grad_y):
        H, W, step = ctx.geom
        B, C, _ = grad_y.shape
        return _image_to_sequences(grad_y.reshape(B, C, H, W), step), None, None, None


class SelectiveScan:
    """cross.py:91-137: a thin autograd Function over the extension; here the extension is libmedscan.so behind
    selective_scan_fn, which is its own autograd Function (so this is a namespace with the same `.apply`)."""

    @staticmethod
    def apply(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
        assert nrows in [1, 2, 3, 4], f"{nrows}"
        with torch.autocast(device_type="cuda", enabled=False):                              # custom_fwd(cast_inputs=float32)
            f = lambda t: t.float() if t is not None else None
            return selective_scan_fn(f(u), f(delta), f(A), f(B), f(C), f(D), None, f(delta_bias), delta_softplus)


def _core(x, x_proj_weight, x_proj_bias, dt_projs_weight, dt_projs_bias, A_logs, Ds, out_norm, nrows, delta_softplus, to_dtype,
          step_size):
    B, _, H, W = x.shape
    N = A_logs.shape[1]
    K, D, R = dt_projs_weight.shape
    if (_LATTICE_KERNEL and x.is_cuda and step_size == 2 and K == 4 and N <= 16 and R <= 32 and delta_softplus
            and H * W < (1 << 20)):
        # the sub-lattices as an addressing mode of the scan kernels: pixel-order operands, no gathers (see the module docstring)
        from .ss2d_fused import _SS2DScan
        with torch.autocast(device_type="cuda", enabled=False):
            Hp, Wp = H + (H & 1), W + (W & 1)
            xc = x.float().permute(0, 2, 3, 1)                                                # (B,H,W,D)
            if (Hp, Wp) != (H, W):
                xc = F.pad(xc, (0, 0, 0, Wp - W, 0, Hp - H))
            xc = xc.contiguous()
            C = R + 2 * N
            proj = F.linear(xc.view(B * Hp * Wp, D), x_proj_weight.float().reshape(4 * C, D),
                            x_proj_bias.float().reshape(4 * C) if x_proj_bias is not None else None).view(B, Hp * Wp, 4, C)
            y = _SS2DScan.apply(xc, proj, None, dt_projs_weight.float(), -torch.exp(A_logs.float()), Ds.float().view(-1),
                                dt_projs_bias.float().view(-1), Hp, Wp, N, R, True)            # (B, Hp*Wp, D), merged
            y = y.view(B, Hp, Wp, D)[:, :H, :W, :]
            y = out_norm(y.reshape(B, H * W, D)).view(B, H, W, -1)
        return y.to(x.dtype) if to_dtype else y
    xs = EfficientScan.apply(x, step_size)                                                   # (B,4,D,L2)
    L = xs.shape[-1]
    x_dbl = torch.einsum("b k d l, k c d -> b k c l", xs, x_proj_weight)
    if x_proj_bias is not None:
        x_dbl = x_dbl + x_proj_bias.view(1, K, -1, 1)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("b k r l, k d r -> b k d l", dts, dt_projs_weight)
    ys = SelectiveScan.apply(xs.reshape(B, -1, L).float(), dts.contiguous().view(B, -1, L).float(), -torch.exp(A_logs.float()),
                             Bs.contiguous().float(), Cs.contiguous().float(), Ds.float(), dt_projs_bias.view(-1).float(),
                             delta_softplus, max(1, nrows)).view(B, K, -1, L)
    y = EfficientMerge.apply(ys, H, W, step_size)                                            # (B,D,H*W)
    y = out_norm(y.transpose(1, 2).contiguous()).view(B, H, W, -1)
    return y.to(x.dtype) if to_dtype else y


def cross_selective_scan(x=None, x_proj_weight=None, x_proj_bias=None, dt_projs_weight=None, dt_projs_bias=None, A_logs=None,
                         Ds=None, out_norm=None, nrows=-1, delta_softplus=True, to_dtype=True, step_size=2):
    """cross.py:265-335 (and the identical cross_selective_scan_new, :193-263): x (B,D,H,W) -> out_norm(merge(scan)) (B,H,W,D)."""
    return _core(x, x_proj_weight, x_proj_bias, dt_projs_weight, dt_projs_bias, A_logs, Ds, out_norm, nrows, delta_softplus,
                 to_dtype, step_size)


cross_selective_scan_new = cross_selective_scan


def cross_selective_scan_cross(x1=None, x2=None, x_proj_weight=None, x_proj_bias=None, dt_projs_weight=None, dt_projs_bias=None,
                               A_logs=None, Ds=None, out_norm=None, nrows=-1, delta_softplus=True, to_dtype=True, step_size=2):
    """cross.py:338-414: the two modalities enter as x1*x2 + x1 + x2."""
    return _core(x1 * x2 + x1 + x2, x_proj_weight, x_proj_bias, dt_projs_weight, dt_projs_bias, A_logs, Ds, out_norm, nrows,
                 delta_softplus, to_dtype, step_size)
