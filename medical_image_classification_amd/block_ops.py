"""torch.autograd wrappers of the HIP kernels around SS2D inside the two-branch block `SS_Conv_SSM`
(MedMamba.py:502-538; C ABI: include/medscan.h):

  layernorm_rows   <- `self.ln_1(right)` on the strided right half of the block input (MedMamba.py:512-515): reads
                      the half in place, writes the dtype the in-projection consumes
  split_halves     <- `input.chunk(2, dim=-1)`: same forward views, but ONE concat in backward instead of two
                      zero-fill + slice-copy + add chains
  block_tail       <- drop_path + `torch.cat((left, x), -1)` + `channel_shuffle(.., 2)` + `+ input`
                      (MedMamba.py:515,534-538,486-499) in one pass; the residual's gradient is `dout` itself
All need CUDA (HIP) tensors and raise RuntimeError otherwise -- no CPU fallback.
"""
import ctypes

import torch

from . import _lib, arena, shadow


def _stream(t):
    return _lib.current_stream_ptr(t.device)


class _LayerNormRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_bf16):
        _lib.require_cuda(x, weight, bias)
        D = x.shape[-1]
        ps = x.stride(-2) if x.dim() > 1 else D
        # rows must be addressable as pixel * ps: fp32, unit channel stride and a uniform pixel stride over all leading dims
        ok = x.dtype == torch.float32 and x.stride(-1) == 1 and ps >= D
        if ok:
            exp = ps
            for d in range(x.dim() - 2, -1, -1):
                if x.shape[d] != 1 and x.stride(d) != exp:
                    ok = False
                    break
                exp *= x.shape[d]
        if not ok:
            # one pass: cast + layout (Tensor.to ignores memory_format when the dtype already matches)
            x = x.contiguous() if x.dtype == torch.float32 else x.to(dtype=torch.float32, memory_format=torch.contiguous_format)
            ps = D
        npix = x.numel() // D
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous()
        out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ms_layernorm_fwd(x.data_ptr(), ps, w.data_ptr(), b.data_ptr(), float(eps), out.data_ptr(),
                                                   int(out_bf16), npix, D, _stream(x)), "ms_layernorm_fwd")
        ctx.save_for_backward(x, w)
        ctx.ps, ctx.eps, ctx.wdtype, ctx.bdtype = ps, float(eps), weight.dtype, bias.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        D = x.shape[-1]
        npix = x.numel() // D
        if dout.dtype not in (torch.float32, torch.bfloat16):
            dout = dout.float()
        dout = dout.contiguous()
        dx = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        dgb = arena.zeros((2, D), x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ms_layernorm_bwd(x.data_ptr(), ctx.ps, w.data_ptr(), ctx.eps, dout.data_ptr(),
                                                   int(dout.dtype == torch.bfloat16), dx.data_ptr(), dgb[0].data_ptr(),
                                                   dgb[1].data_ptr(), npix, D, _stream(x)), "ms_layernorm_bwd")
        return dx, dgb[0].to(ctx.wdtype), dgb[1].to(ctx.bdtype), None, None


class _LayerNormTaps(torch.autograd.Function):
    """LayerNorm(4C) over the 2 x 2 tap concatenation of a (B, H, W, C) fp32 tensor: PatchMerging2D's gather and norm
    (MedMamba.py:196-205) in ONE pass each way (ms_layernorm_taps_*: the gather is an addressing mode of the LayerNorm kernel, the
    scatter of its dx store) instead of a permuting copy + LayerNorm forward and LayerNorm + permuting copy backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_bf16):
        B, H, W, C = x.shape
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous()
        out = torch.empty((B, H // 2, W // 2, 4 * C), device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ms_layernorm_taps_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), float(eps), out.data_ptr(), int(out_bf16),
                                                        B, H, W, C, _stream(x)), "ms_layernorm_taps_fwd")
        ctx.save_for_backward(x, w)
        ctx.eps, ctx.wdtype, ctx.bdtype = float(eps), weight.dtype, bias.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        B, H, W, C = x.shape
        if dout.dtype not in (torch.float32, torch.bfloat16):
            dout = dout.float()
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        dgb = arena.zeros((2, 4 * C), x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ms_layernorm_taps_bwd(x.data_ptr(), w.data_ptr(), ctx.eps, dout.data_ptr(), int(dout.dtype == torch.bfloat16),
                                                        dx.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(), B, H, W, C, _stream(x)),
                       "ms_layernorm_taps_bwd")
        return dx, dgb[0].to(ctx.wdtype), dgb[1].to(ctx.bdtype), None, None


def layernorm_taps_ok(x, norm):
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0
            and x.shape[3] % 4 == 0 and 4 * x.shape[3] <= 2048 and x.data_ptr() % 16 == 0 and type(norm) is torch.nn.LayerNorm
            and norm.elementwise_affine and norm.bias is not None and tuple(norm.normalized_shape) == (4 * x.shape[3],))


def layernorm_taps(x, norm, out_bf16=None):
    """`norm(cat of the four 2 x 2 taps of x)` -> (B, H/2, W/2, 4C); out_bf16=None: bf16 under bf16 autocast."""
    if out_bf16 is None:
        out_bf16 = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
    with torch.autocast(device_type="cuda", enabled=False):
        return _LayerNormTaps.apply(x, norm.weight, norm.bias, norm.eps, bool(out_bf16))


def layernorm_rows(x, weight, bias, eps, out_bf16=None):
    """LayerNorm over the last axis of a (.., D) fp32 tensor that may be a channel slice of a wider tensor.
    out_bf16=None: bf16 when CUDA autocast to bf16 is on (what the next linear would cast to), fp32 otherwise."""
    if out_bf16 is None:
        out_bf16 = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
    return _LayerNormRows.apply(x, weight, bias, eps, bool(out_bf16))


class BlockFrame:
    """Links the two ends of one block's residual frame, `split_halves(input, frame)` and `block_tail(.., input, .., frame=frame)`:
    the tail's backward parks the residual edge's gradient (`dout` itself) here instead of returning it, and the split's backward
    -- which autograd runs after both branches -- emits d_input = dout + cat(d_left, d_right) in one pass (ms_block_head_bwd)
    instead of a concat followed by autograd's accumulation add."""
    __slots__ = ("dout",)

    def __init__(self):
        self.dout = None


class _SplitHalves(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, frame):
        half = x.shape[-1] // 2
        ctx.dtype, ctx.frame = x.dtype, frame
        return x[..., :half], x[..., half:]

    @staticmethod
    def backward(ctx, dl, dr):
        frame = ctx.frame
        dout = frame.dout if frame is not None else None
        if dout is None:
            return torch.cat((dl.to(ctx.dtype), dr.to(ctx.dtype)), dim=-1), None
        frame.dout = None
        C = dout.shape[-1]
        ok = (ctx.dtype == torch.float32 and C % 8 == 0 and dl.shape[-1] * 2 == C and dl.shape[:-1] == dout.shape[:-1] == dr.shape[:-1]
              and dl.dtype in (torch.float32, torch.bfloat16) and dr.dtype in (torch.float32, torch.bfloat16))
        if not ok:
            return torch.cat((dl.to(ctx.dtype), dr.to(ctx.dtype)), dim=-1) + dout.to(ctx.dtype), None
        dl, dr = dl.contiguous(), dr.contiguous()
        dinp = torch.empty_like(dout)
        with _lib.on_device(dout.device):
            _lib.check(_lib.lib().ms_block_head_bwd(dout.data_ptr(), dl.data_ptr(), int(dl.dtype == torch.bfloat16), dr.data_ptr(),
                                                    int(dr.dtype == torch.bfloat16), dinp.data_ptr(), dout.numel() // C, C,
                                                    _stream(dout)), "ms_block_head_bwd")
        return dinp, None


def split_halves(x, frame=None):
    """`x.chunk(2, dim=-1)` (views) whose backward is a single concat -- or, with the `BlockFrame` that is also given to this
    block's `block_tail`, one pass that adds the residual edge's gradient as well."""
    return _SplitHalves.apply(x, frame)


class _BlockTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, left, x, inp, scale, frame, left_relu):
        """frame: BlockFrame shared with split_halves(inp, frame) or None.  left_relu: `left` is the output of the ReLU that ends
        the conv branch and its producer expects the gradient w.r.t. that ReLU's INPUT (conv1x1_relu(.., premasked=True))."""
        _lib.require_cuda(left, x, inp)
        B, H, W, C = inp.shape
        half = C // 2
        if tuple(left.shape) != (B, H, W, half) or tuple(x.shape) != (B, H, W, half):
            raise RuntimeError(f"block_tail: halves must be {(B, H, W, half)}, got {tuple(left.shape)} and {tuple(x.shape)}")
        if C % 4 != 0:
            raise RuntimeError("block_tail: channel count must be a multiple of 4")
        norm = lambda t: (t if t.dtype in (torch.float32, torch.bfloat16) else t.float()).contiguous()
        left, x = norm(left), norm(x)
        inp = inp.float().contiguous()
        sc = scale.detach().float().contiguous().view(-1) if scale is not None else None
        if sc is not None and sc.numel() != B:
            raise RuntimeError("block_tail: sample_scale must have one entry per sample")
        out = torch.empty_like(inp)
        with _lib.on_device(inp.device):
            _lib.check(_lib.lib().ms_block_tail_fwd(left.data_ptr(), int(left.dtype == torch.bfloat16), x.data_ptr(),
                                                    int(x.dtype == torch.bfloat16), inp.data_ptr(),
                                                    sc.data_ptr() if sc is not None else None, out.data_ptr(),
                                                    B * H * W, H * W, C, _stream(inp)), "ms_block_tail_fwd")
        ctx.scale, ctx.ldt, ctx.xdt, ctx.frame = sc, left.dtype, x.dtype, frame
        ctx.left = left if left_relu else None          # (kept by the ReLU's producer anyway: no extra memory)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.float().contiguous()
        B, H, W, C = dout.shape
        dl = torch.empty((B, H, W, C // 2), device=dout.device, dtype=ctx.ldt)
        dx = torch.empty((B, H, W, C // 2), device=dout.device, dtype=ctx.xdt)
        sc = ctx.scale
        with _lib.on_device(dout.device):
            if ctx.left is not None:
                _lib.check(_lib.lib().ms_block_tail_bwd_relu(dout.data_ptr(), sc.data_ptr() if sc is not None else None,
                                                             ctx.left.data_ptr(), dl.data_ptr(), int(ctx.ldt == torch.bfloat16),
                                                             dx.data_ptr(), int(ctx.xdt == torch.bfloat16), B * H * W, H * W, C,
                                                             _stream(dout)), "ms_block_tail_bwd_relu")
            else:
                _lib.check(_lib.lib().ms_block_tail_bwd(dout.data_ptr(), sc.data_ptr() if sc is not None else None, dl.data_ptr(),
                                                        int(ctx.ldt == torch.bfloat16), dx.data_ptr(), int(ctx.xdt == torch.bfloat16),
                                                        B * H * W, H * W, C, _stream(dout)), "ms_block_tail_bwd")
        if ctx.frame is not None and ctx.needs_input_grad[2]:
            ctx.frame.dout = dout                        # added to cat(d_left, d_right) by the split's backward, in one pass
            return dl, dx, None, None, None, None
        return dl, dx, dout, None, None, None


def block_tail(left, x, inp, sample_scale=None, frame=None, left_relu=False):
    """channel_shuffle(cat(left, sample_scale * x), groups=2) + inp for channel-last (B,H,W,C/2) halves and a (B,H,W,C) input."""
    return _BlockTail.apply(left, x, inp, sample_scale, frame, left_relu)


# ---- training-mode BatchNorm2d (+ ReLU) of the conv branch, channels_last ---------------------------------------------
_BN_SCRATCH = {}       # (device, stream, C) -> workspace for the per-workgroup partial sums (reused in stream order)


def _bn_scratch(device, C):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, C)
    buf = _BN_SCRATCH.get(key)
    if buf is None:
        buf = _BN_SCRATCH[key] = torch.empty(_lib.lib().ms_bn_scratch_floats(C), device=device, dtype=torch.float32)
    return buf


def _nhwc_pixel_stride(t):
    """Pixel stride of an NCHW-shaped tensor whose memory is rows of C channels (unit channel stride, uniform stride
    between pixels: channels_last, or a channel slice of a wider channels_last tensor); None if it is anything else."""
    if t.dim() != 4 or t.dtype not in (torch.float32, torch.bfloat16):
        return None
    B, C, H, W = t.shape
    ps = t.stride(3)
    if t.stride(1) != 1 or ps < C or t.stride(2) != W * ps or (B > 1 and t.stride(0) != H * W * ps):
        return None
    return ps


def _nhwc_ok(t):
    return _nhwc_pixel_stride(t) == t.shape[1]


class _BatchNormReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, momentum, eps, relu, out_bf16, shift):
        """shift: bias of the convolution in front (or None) -- see `conv_branch`; its gradient is identically zero."""
        B, C, H, W = x.shape
        npix = B * H * W
        ps = _nhwc_pixel_stride(x)
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous()
        y = torch.empty((B, C, H, W), device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32,
                        memory_format=torch.channels_last)
        save = torch.empty((2, C), device=x.device, dtype=torch.float32)
        with _lib.on_device(x.device):
            sh = shift.detach().float().contiguous() if shift is not None else None
            _lib.check(_lib.lib().ms_bn_relu_nhwc_fwd(
                x.data_ptr(), int(x.dtype == torch.bfloat16), ps, sh.data_ptr() if sh is not None else None, w.data_ptr(),
                b.data_ptr(), running_mean.data_ptr(),
                running_var.data_ptr(), nbt.data_ptr() if nbt is not None else None, float(momentum), float(eps), int(relu),
                y.data_ptr(), int(out_bf16), save[0].data_ptr(), save[1].data_ptr(), _bn_scratch(x.device, C).data_ptr(),
                npix, C, _stream(x)), "ms_bn_relu_nhwc_fwd")
        ctx.save_for_backward(x, w, b, save)
        ctx.relu, ctx.wdtype, ctx.bdtype, ctx.ps = bool(relu), weight.dtype, bias.dtype, ps
        ctx.shift_like = shift if shift is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, save = ctx.saved_tensors
        B, C, H, W = x.shape
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty(dy.shape, device=dy.device, dtype=x.dtype, memory_format=torch.channels_last)   # written in x's dtype: no cast pass
        dgb = torch.empty((2, C), device=x.device, dtype=torch.float32)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ms_bn_relu_nhwc_bwd(
                x.data_ptr(), int(x.dtype == torch.bfloat16), ctx.ps, dy.data_ptr(), int(dy.dtype == torch.bfloat16), w.data_ptr(),
                b.data_ptr(), save[0].data_ptr(), save[1].data_ptr(), int(ctx.relu), dx.data_ptr(), int(dx.dtype == torch.bfloat16),
                dgb[0].data_ptr(), dgb[1].data_ptr(), _bn_scratch(x.device, C).data_ptr(), B * H * W, C, _stream(x)), "ms_bn_relu_nhwc_bwd")
        dshift = arena.zeros_like(ctx.shift_like) if ctx.shift_like is not None else None      # d/d(shift) of BN(x + shift) == 0
        return dx, dgb[0].to(ctx.wdtype), dgb[1].to(ctx.bdtype), None, None, None, None, None, None, None, dshift


def batchnorm_relu(bn, x, relu, shift=None):
    """`relu(bn(x + shift))` (or `bn(x + shift)`; shift = per-channel constant or None) for an nn.BatchNorm2d in TRAINING mode on a channels_last CUDA tensor, through
    ms_bn_relu_nhwc_* (batch statistics, running-statistics update and num_batches_tracked as torch does).  Anything else
    (eval mode, no affine / no running stats, cumulative momentum, other layouts) goes through the module itself."""
    if not (bn.training and x.is_cuda and _nhwc_pixel_stride(x) is not None and bn.affine and bn.track_running_stats and bn.momentum is not None
            and type(bn) is torch.nn.BatchNorm2d and bn.running_mean is not None):
        x = x.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else x
        y = bn(x if shift is None else x + shift.view(1, -1, 1, 1).to(x.dtype))
        return torch.relu(y) if relu else y
    out_bf16 = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)
    return _BatchNormReLU.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum,
                                bn.eps, relu, out_bf16, shift)


class _ConvShadow(torch.autograd.Function):
    """conv2d (groups 1, no bias) of a bf16 channels_last activation with the cached bf16 channels_last copy of an fp32 weight
    (shadow.bf16): no per-use weight cast / layout copy in forward; the weight gradient leaves as ONE fp32 contiguous tensor
    (autograd's chain was: bf16 channels_last dW -> strided cast to fp32 -> a second, layout-fixing copy inside AccumulateGrad)."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, dilation):
        wb = shadow.bf16(weight, conv=True)
        ctx.direct = _conv3x3_direct_ok(x, weight, stride, padding, dilation)
        if ctx.direct:
            y = _conv3x3_direct(x, wb)                    # ms_conv3x3_nhwc_bf16: csrc/conv3x3.hip
        else:
            y = torch.ops.aten.convolution(x, wb, None, stride, padding, dilation, False, [0, 0], 1)
        ctx.save_for_backward(x, wb)
        ctx.geom = (stride, padding, dilation)
        ctx.weight_ref = weight
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        shadow.invalidate(x.device)              # a backward pass is under way: the weights are about to change
        stride, padding, dilation = ctx.geom
        dy = dy.contiguous(memory_format=torch.channels_last)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        need_dx = ctx.needs_input_grad[0]
        dx_direct = None
        if ctx.direct and need_dx:
            # dx = conv3x3(dy, w') with the flipped / transposed weight copy (same kernel as the forward)
            dx_direct = _conv3x3_direct(dy, shadow.bf16(ctx.weight_ref, conv="flip", in_backward=True))
            need_dx = False
        need_dw = ctx.needs_input_grad[1]
        dw_direct = None
        if ctx.direct and need_dw and _CONV_WGRAD:
            dw_direct = _conv3x3_wgrad(x, dy, wb.shape)  # fp32 (Co, Ci, 3, 3), written: no zero-fill, no cast pass
            need_dw = False
        dx = dw = None
        if need_dx or need_dw:
            dx, dw, _ = torch.ops.aten.convolution_backward(dy, x, wb, None, stride, padding, dilation, False, [0, 0], 1,
                                                            [need_dx, need_dw, False])
        if dx_direct is not None:
            dx = dx_direct
        if dw_direct is not None:
            return dx, dw_direct, None, None, None
        if dw is not None:
            dw = dw.to(dtype=torch.float32, memory_format=torch.contiguous_format)
        return dx, dw, None, None, None


_CONV_DIRECT = __import__("os").environ.get("MEDSCAN_CONV3X3", "1") == "1"
_CONV_DIRECT_MIN_PIXELS = int(__import__("os").environ.get("MEDSCAN_CONV3X3_MIN_PIXELS", "49"))       # per image; 7x7 maps (stage 3) included since the pipelined kernel: 42 vs 53 us


def _conv3x3_direct_ok(x, weight, stride, padding, dilation):
    return (_CONV_DIRECT and x.is_cuda and x.dtype == torch.bfloat16 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)
            and list(stride) == [1, 1] and list(padding) == [1, 1] and list(dilation) == [1, 1]
            and weight.shape[0] % 16 == 0 and weight.shape[1] % 16 == 0 and x.shape[2] * x.shape[3] >= _CONV_DIRECT_MIN_PIXELS
            and x.is_contiguous(memory_format=torch.channels_last))


# the hand-written weight-gradient kernel (ms_conv3x3_wgrad: persistent MFMA workgroups + a coalesced partial-block sum) replaces
# MIOpen's bf16 weight-gradient kernels and the zero-fill / cast helper launches around them: cold 52-59 us per call against 49-62 us,
# MedMamba-T step 20.28 -> 20.14 ms and ~100 fewer launches per step.  MEDSCAN_CONV3X3_WGRAD=0 goes back to MIOpen.
_CONV_WGRAD = __import__("os").environ.get("MEDSCAN_CONV3X3_WGRAD", "1") == "1"


def _conv3x3_wgrad(x, dy, wshape):
    """Weight gradient of the 3x3 / stride 1 / padding 1 convolution from bf16 channels_last x and dy: fp32 (Co, Ci, 3, 3)."""
    B, Ci, H, W = x.shape
    Co = dy.shape[1]
    lib = _lib.lib()
    ns = lib.ms_conv3x3_wgrad_scratch_floats(B, H, W, Ci, Co)
    scratch = torch.empty(ns, device=x.device, dtype=torch.float32)
    dw = torch.empty((Co, Ci, 3, 3), device=x.device, dtype=torch.float32)
    with _lib.on_device(x.device):
        _lib.check(lib.ms_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), scratch.data_ptr(), ns, B, H, W, Ci, Co, _stream(x)),
                   "ms_conv3x3_wgrad")
    return dw


def _conv3x3_direct(x, w_nhwc):
    """3x3 / stride 1 / padding 1 convolution of a bf16 channels_last activation with a bf16 weight in (Co, 3, 3, Ci) memory."""
    B, Ci, H, W = x.shape
    Co = w_nhwc.numel() // (9 * Ci)
    y = torch.empty((B, Co, H, W), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().ms_conv3x3_nhwc_bf16(x.data_ptr(), w_nhwc.data_ptr(), y.data_ptr(), B, H, W, Ci, Co, _stream(x)),
                   "ms_conv3x3_nhwc_bf16")
    return y


def _conv2d(conv, x):
    """conv(x) without bias: through the weight's cached bf16 copy under bf16 autocast (see _ConvShadow), F.conv2d otherwise."""
    if (x.is_cuda and x.dtype == torch.bfloat16 and conv.weight.dtype == torch.float32 and conv.groups == 1
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and x.is_contiguous(memory_format=torch.channels_last)):
        with torch.autocast(device_type="cuda", enabled=False):
            return _ConvShadow.apply(x, conv.weight, list(conv.stride), list(conv.padding), list(conv.dilation))
    return torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


class _Conv1x1ReLU(torch.autograd.Function):
    """relu(conv1x1(x) + bias) on a channels_last activation = one GEMM over the pixel rows with a bias + ReLU epilogue
    (ms_gemm_bf16_bias_act; MedMamba.py:525-526).  Backward: dz = dy * [y > 0] (or dy itself when the consumer -- block_tail with
    left_relu -- has already applied the mask), dx = dz W (ms_gemm_bf16), dW = dz^T x (split-K inside the kernel, fp32),
    dbias = column sums of dz.  The fp32 master weight is read directly (rounded to bf16 while a tile is staged)."""

    @staticmethod
    def forward(ctx, x, weight, bias, premasked):
        from .gemm_ops import gemm
        B, C, H, W = x.shape
        Co = weight.shape[0]
        xm = x.permute(0, 2, 3, 1).reshape(B * H * W, C)             # a view: x is channels_last
        w2 = weight.detach().view(Co, C)
        y = gemm(xm, w2, out_dtype=torch.bfloat16, bias=bias.detach() if bias is not None else None, relu=True)
        ctx.save_for_backward(xm, w2, y)
        ctx.premasked, ctx.shape, ctx.has_bias = premasked, (B, C, H, W), bias is not None
        return y.view(B, H, W, Co).permute(0, 3, 1, 2)                # NCHW-shaped, channels_last memory

    @staticmethod
    def backward(ctx, dy):
        from .gemm_ops import gemm, weight_grad
        xm, w2, y = ctx.saved_tensors
        B, C, H, W = ctx.shape
        Co = w2.shape[0]
        dz = dy.permute(0, 2, 3, 1).reshape(B * H * W, Co)
        if not ctx.premasked:
            dz = torch.ops.aten.threshold_backward(dz.contiguous(), y, 0)
        if dz.dtype not in (torch.bfloat16, torch.float32) or not dz.is_contiguous():
            dz = dz.contiguous().float()
        dx = gemm(dz, w2, b_trans=True, out_dtype=xm.dtype) if ctx.needs_input_grad[0] else None
        from .gemm_ops import wgrad_bias_ok
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and want_db and wgrad_bias_ok(dz, xm):
            db = arena.zeros((Co,), dz.device)                # the column sums of dz ride on the weight-gradient launch
            dw = weight_grad(dz, xm, dbias=db)
        else:
            dw = weight_grad(dz, xm) if ctx.needs_input_grad[1] else None
            db = dz.sum(dim=0, dtype=torch.float32) if want_db else None
        return (dx.view(B, H, W, C).permute(0, 3, 1, 2) if dx is not None else None,
                dw.view(Co, C, 1, 1) if dw is not None else None, db, None)


def conv1x1_relu(conv, x, premasked=False):
    """relu(conv(x)) for a 1x1 nn.Conv2d on a bf16 channels_last CUDA activation through the MFMA GEMM; None if not applicable."""
    C = conv.in_channels
    if not (type(conv) is torch.nn.Conv2d and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.groups == 1 and x.is_cuda and x.dtype == torch.bfloat16 and conv.weight.dtype == torch.float32
            and C % 8 == 0 and conv.out_channels % 8 == 0 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return None
    with torch.autocast(device_type="cuda", enabled=False):
        return _Conv1x1ReLU.apply(x, conv.weight, conv.bias, premasked)


# ---- conv3x3 -> BN -> ReLU -> conv3x3 -> BN -> ReLU with the two BatchNorms folded into the convolutions (csrc/conv3x3.hip, ABI v9) ----------
_BN_FOLD = __import__("os").environ.get("MEDSCAN_CONV_BN_FOLD", "1") == "1"
FOLD_CALLS = 0          # how often conv_branch took the folded form (tests)


def _bn_fold_desc(sums, gamma, beta, shift, bn, save):
    d = _lib.MsBnFold()
    d.sums, d.gamma, d.beta = sums.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    d.shift = shift.data_ptr() if shift is not None else None
    d.running_mean, d.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
    d.num_batches_tracked = bn.num_batches_tracked.data_ptr() if bn.num_batches_tracked is not None else None
    d.save_mean, d.save_rstd = save[0].data_ptr(), save[1].data_ptr()
    d.momentum, d.eps = float(bn.momentum), float(bn.eps)
    return d


def _bn_bwd_desc(xpre, ps, gamma, beta, save, relu, sums):
    d = _lib.MsBnBwd()
    d.x_pre, d.x_pre_is_f32, d.x_pre_pixel_stride = xpre.data_ptr(), int(xpre.dtype == torch.float32), ps
    d.gamma, d.beta, d.save_mean, d.save_rstd = gamma.data_ptr(), beta.data_ptr(), save[0].data_ptr(), save[1].data_ptr()
    d.relu, d.sums = int(relu), sums.data_ptr()
    return d


class _ConvBnConvBn(torch.autograd.Function):
    """The whole conv branch `relu(conv1x1(relu(bn2(conv2(relu(bn1(conv1(bn0(x)))))))))` of SS_Conv_SSM (`conv33conv33conv11`,
    MedMamba.py:517-527) for training-mode BatchNorms on a channels_last activation, as ONE autograd node.
    Forward: bn0 as statistics / finalize / apply (its input comes from outside); conv1 accumulates the statistics of its output in its
    epilogue; conv2 normalises + rectifies conv1's output while it stages its input tiles (writing the normalised activation `x1` as a
    side output for the backward) and accumulates the statistics of ITS output; one apply pass produces x2; the 1x1 convolution + bias +
    ReLU is ms_gemm_bf16_bias_act.  (Was 2 convolutions + 3 x (statistics, finalize, apply) + the GEMM.)
    Backward: the reduce pass of every BatchNorm rides in the epilogue of the launch that produces its incoming gradient -- bn2's in the
    1x1 convolution's input-gradient GEMM (ms_gemm_bf16_bnbwd), bn1's and bn0's in the input-gradient launches of conv2 and conv1
    (`BRED`); their apply passes read the replica rows and write dgamma / dbeta.
    The 3x3 convolutions' biases never touch the activations (a constant in front of a mean subtraction): they enter the running means
    only and get exact zero gradients."""

    @staticmethod
    def forward(ctx, x, g0, be0, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, bn0, bn1, bn2, premasked):
        # w3 / b3: the 1x1 convolution + ReLU that ends the branch (MedMamba.py:525-526) on ms_gemm_bf16_bias_act; in the backward its input
        # gradient launch carries bn2's reduce (ms_gemm_bf16_bnbwd).  premasked: the incoming gradient is already masked by that ReLU.
        B, C, H, W = x.shape
        npix = B * H * W
        lib = _lib.lib()
        ps = _nhwc_pixel_stride(x)
        wb1, wb2 = shadow.bf16(w1, conv=True), shadow.bf16(w2, conv=True)
        f32 = lambda t: t.detach().float().contiguous()
        g0f, be0f, g1f, be1f, g2f, be2f = f32(g0), f32(be0), f32(g1), f32(be1), f32(g2), f32(be2)
        b1f, b2f = (f32(b1) if b1 is not None else None), (f32(b2) if b2 is not None else None)
        nf = (2 * _lib.BN_REPLICAS + 1) * C
        sums = torch.zeros(2 * nf, device=x.device, dtype=torch.float32)
        save = torch.empty((6, C), device=x.device, dtype=torch.float32)
        new = lambda: torch.empty((B, C, H, W), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
        x0, y1, x1, y2, x2 = new(), new(), new(), new(), new()
        d1 = _bn_fold_desc(sums[:nf], g1f, be1f, b1f, bn1, save[2:4])
        d2 = _bn_fold_desc(sums[nf:], g2f, be2f, b2f, bn2, save[4:6])
        st = _stream(x)
        with _lib.on_device(x.device):
            _lib.check(lib.ms_bn_relu_nhwc_fwd(x.data_ptr(), int(x.dtype == torch.bfloat16), ps, None, g0f.data_ptr(), be0f.data_ptr(),
                                               bn0.running_mean.data_ptr(), bn0.running_var.data_ptr(),
                                               bn0.num_batches_tracked.data_ptr() if bn0.num_batches_tracked is not None else None,
                                               float(bn0.momentum), float(bn0.eps), 0, x0.data_ptr(), 1, save[0].data_ptr(), save[1].data_ptr(),
                                               _bn_scratch(x.device, C).data_ptr(), npix, C, st), "ms_bn_relu_nhwc_fwd")
            _lib.check(lib.ms_conv3x3_bn_nhwc_bf16(x0.data_ptr(), wb1.data_ptr(), y1.data_ptr(), B, H, W, C, C, None, None, ctypes.byref(d1), st),
                       "ms_conv3x3_bn_nhwc_bf16")
            _lib.check(lib.ms_conv3x3_bn_nhwc_bf16(y1.data_ptr(), wb2.data_ptr(), y2.data_ptr(), B, H, W, C, C, ctypes.byref(d1), x1.data_ptr(),
                                                   ctypes.byref(d2), st), "ms_conv3x3_bn_nhwc_bf16")
            _lib.check(lib.ms_bn_apply_sums_nhwc(y2.data_ptr(), ctypes.byref(d2), 1, x2.data_ptr(), npix, C, st), "ms_bn_apply_sums_nhwc")
        from .gemm_ops import gemm
        Co = w3.shape[0]
        w3m = w3.detach().view(Co, C)
        out = gemm(x2.permute(0, 2, 3, 1).reshape(npix, C), w3m, out_dtype=torch.bfloat16, bias=b3.detach() if b3 is not None else None, relu=True)
        ctx.save_for_backward(x, x0, y1, x1, y2, x2, out, g0f, be0f, g1f, be1f, g2f, be2f, save)
        ctx.w1, ctx.w2, ctx.w3, ctx.b1, ctx.b2, ctx.b3, ctx.ps, ctx.premasked = w1, w2, w3m, b1, b2, b3, ps, premasked
        ctx.dtypes = (g0.dtype, be0.dtype, g1.dtype, be1.dtype, g2.dtype, be2.dtype)
        return out.view(B, H, W, Co).permute(0, 3, 1, 2)                # NCHW-shaped, channels_last memory

    @staticmethod
    def backward(ctx, dy):
        from .gemm_ops import weight_grad, wgrad_bias_ok
        x, x0, y1, x1, y2, x2, out, g0f, be0f, g1f, be1f, g2f, be2f, save = ctx.saved_tensors
        B, C, H, W = x.shape
        npix = B * H * W
        dev = x.device
        shadow.invalidate(dev)
        lib = _lib.lib()
        new = lambda dt=torch.bfloat16: torch.empty((B, C, H, W), device=dev, dtype=dt, memory_format=torch.channels_last)
        st = _stream(dy)
        # the 1x1 convolution + ReLU: dz = dy * [out > 0] (or dy itself when the consumer has applied the mask), dW3 = dz^T x2 with the
        # column sums of dz (the bias gradient) on the same launch, dx2 = dz W3 with bn2's reduce in its epilogue
        w3m = ctx.w3
        Co = w3m.shape[0]
        dz = dy.permute(0, 2, 3, 1).reshape(npix, Co)
        if not ctx.premasked:
            dz = torch.ops.aten.threshold_backward(dz.contiguous(), out, 0)
        if dz.dtype not in (torch.bfloat16, torch.float32) or not dz.is_contiguous():
            dz = dz.contiguous().float()
        x2m = x2.permute(0, 2, 3, 1).reshape(npix, C)
        want_db = ctx.b3 is not None and ctx.needs_input_grad[12]
        dw3 = db3 = None
        if ctx.needs_input_grad[11]:
            if want_db and wgrad_bias_ok(dz, x2m):
                db3 = arena.zeros((Co,), dev)
                dw3 = weight_grad(dz, x2m, dbias=db3)
            else:
                dw3 = weight_grad(dz, x2m)
        if want_db and db3 is None:
            db3 = dz.sum(dim=0, dtype=torch.float32)
        bsums = arena.zeros((3, 2 * _lib.BN_REPLICAS * C), dev)
        dgb = torch.empty((6, C), device=dev, dtype=torch.float32)
        r2 = _bn_bwd_desc(y2, C, g2f, be2f, save[4:6], True, bsums[2])
        dx2, dy2 = new(), new()
        with _lib.on_device(dev):
            _lib.check(lib.ms_gemm_bf16_bnbwd(dz.data_ptr(), int(dz.dtype == torch.float32), dz.stride(0), w3m.data_ptr(), int(w3m.dtype == torch.float32),
                                              w3m.stride(0), dx2.data_ptr(), 1, C, npix, C, Co, ctypes.byref(r2), st), "ms_gemm_bf16_bnbwd")
            _lib.check(lib.ms_bn_bwd_apply_sums_nhwc(ctypes.byref(r2), dx2.data_ptr(), dy2.data_ptr(), 1, dgb[4].data_ptr(), dgb[5].data_ptr(), npix, C,
                                                     st), "ms_bn_bwd_apply_sums_nhwc")
        dw2 = _conv3x3_wgrad(x1, dy2, ctx.w2.shape) if ctx.needs_input_grad[7] else None
        # conv2's input gradient with bn1's reduce in its epilogue, then bn1's apply from the rows
        r1 = _bn_bwd_desc(y1, C, g1f, be1f, save[2:4], True, bsums[0])
        dx1, dy1 = new(), new()
        wf2, wf1 = shadow.bf16(ctx.w2, conv="flip", in_backward=True), shadow.bf16(ctx.w1, conv="flip", in_backward=True)
        with _lib.on_device(dev):
            _lib.check(lib.ms_conv3x3_bnbwd_nhwc_bf16(dy2.data_ptr(), wf2.data_ptr(), dx1.data_ptr(), B, H, W, C, C, ctypes.byref(r1), st),
                       "ms_conv3x3_bnbwd_nhwc_bf16")
            _lib.check(lib.ms_bn_bwd_apply_sums_nhwc(ctypes.byref(r1), dx1.data_ptr(), dy1.data_ptr(), 1, dgb[0].data_ptr(), dgb[1].data_ptr(), npix, C,
                                                     st), "ms_bn_bwd_apply_sums_nhwc")
        dw1 = _conv3x3_wgrad(x0, dy1, ctx.w1.shape) if ctx.needs_input_grad[3] else None
        # conv1's input gradient with bn0's reduce (no ReLU; pre-normalisation input = the block's strided left half), then bn0's apply
        dxin = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            r0 = _bn_bwd_desc(x, ctx.ps, g0f, be0f, save[0:2], False, bsums[1])
            dx0 = new()
            dxin = new(x.dtype)                                         # written in x's dtype: no cast pass
            with _lib.on_device(dev):
                _lib.check(lib.ms_conv3x3_bnbwd_nhwc_bf16(dy1.data_ptr(), wf1.data_ptr(), dx0.data_ptr(), B, H, W, C, C, ctypes.byref(r0), st),
                           "ms_conv3x3_bnbwd_nhwc_bf16")
                _lib.check(lib.ms_bn_bwd_apply_sums_nhwc(ctypes.byref(r0), dx0.data_ptr(), dxin.data_ptr(), int(dxin.dtype == torch.bfloat16),
                                                         dgb[2].data_ptr(), dgb[3].data_ptr(), npix, C, st), "ms_bn_bwd_apply_sums_nhwc")
        t0, t1, t2, t3, t4, t5 = ctx.dtypes
        zb = lambda b: arena.zeros_like(b) if b is not None else None           # d/d(conv bias) of BN(conv + bias) == 0
        return (dxin, dgb[2].to(t0), dgb[3].to(t1), dw1, zb(ctx.b1), dgb[0].to(t2), dgb[1].to(t3), dw2, zb(ctx.b2), dgb[4].to(t4), dgb[5].to(t5),
                dw3.view(Co, C, 1, 1) if dw3 is not None else None, db3, None, None, None, None)


def _conv_bn_fold_ok(mods, x):
    """The conv branch as `_ConvBnConvBn`: training-mode BatchNorms with affine + running statistics, bias-or-not 3x3 / stride 1 / padding 1
    convolutions of equal width (a multiple of 16, <= 512), a channels_last-addressable CUDA input, bf16 autocast."""
    bn0, c1, bn1, c2, bn2, c3 = mods[0], mods[1], mods[2], mods[4], mods[5], mods[7]
    def bn_ok(bn):
        return (bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None and type(bn) is torch.nn.BatchNorm2d
                and bn.running_mean is not None and bn.weight.dtype == torch.float32)
    def conv_ok(c):
        return (type(c) is torch.nn.Conv2d and c.padding_mode == "zeros" and c.groups == 1 and c.weight.dtype == torch.float32
                and tuple(c.kernel_size) == (3, 3) and tuple(c.stride) == (1, 1) and tuple(c.padding) == (1, 1) and tuple(c.dilation) == (1, 1)
                and c.in_channels == c.out_channels == x.shape[1])
    C = x.shape[1] if x.dim() == 4 else 0
    ps = _nhwc_pixel_stride(x) if x.dim() == 4 else None
    return (_BN_FOLD and _CONV_WGRAD and _CONV_DIRECT and x.is_cuda and ps is not None and ps % 4 == 0 and x.data_ptr() % 16 == 0 and C % 16 == 0
            and 16 <= C <= 512 and x.shape[2] * x.shape[3] >= _CONV_DIRECT_MIN_PIXELS and bn_ok(bn0) and bn_ok(bn1) and bn_ok(bn2)
            and conv_ok(c1) and conv_ok(c2) and type(c3) is torch.nn.Conv2d and c3.kernel_size == (1, 1) and c3.stride == (1, 1)
            and c3.padding == (0, 0) and c3.groups == 1 and c3.in_channels == C and c3.out_channels % 8 == 0 and c3.weight.dtype == torch.float32
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)


def conv_branch(seq, x, premasked_out=False):
    """The conv branch of SS_Conv_SSM (`self.conv33conv33conv11`, MedMamba.py:517-527) applied module by module with each
    BatchNorm2d (+ following ReLU) fused: BN -> conv3x3 -> BN+ReLU -> conv3x3 -> BN+ReLU -> conv1x1 -> ReLU.  Falls back to
    `seq(x)` when the Sequential is not that exact pattern.
    premasked_out=True returns `(y, masked)`: when `masked`, the backward of `y` expects the gradient w.r.t. the last ReLU's INPUT
    (its consumer applies the mask: block_tail(.., left_relu=True))."""
    mods = list(seq)
    kinds = [type(m) for m in mods]
    nn = torch.nn
    if kinds != [nn.BatchNorm2d, nn.Conv2d, nn.BatchNorm2d, nn.ReLU, nn.Conv2d, nn.BatchNorm2d, nn.ReLU, nn.Conv2d, nn.ReLU]:
        y = seq(x.contiguous(memory_format=torch.channels_last))
        return (y, False) if premasked_out else y
    if _conv_bn_fold_ok(mods, x):
        global FOLD_CALLS
        FOLD_CALLS += 1
        with torch.autocast(device_type="cuda", enabled=False):
            y = _ConvBnConvBn.apply(x, mods[0].weight, mods[0].bias, mods[1].weight, mods[1].bias, mods[2].weight, mods[2].bias,
                                    mods[4].weight, mods[4].bias, mods[5].weight, mods[5].bias, mods[7].weight, mods[7].bias,
                                    mods[0], mods[2], mods[5], bool(premasked_out))
        return (y, True) if premasked_out else y
    else:
        x = batchnorm_relu(mods[0], x, False)
        x = _conv_then_bn(mods[1], mods[2], x)
        x = _conv_then_bn(mods[4], mods[5], x)
    y = conv1x1_relu(mods[7], x, premasked=premasked_out)
    if y is not None:
        return (y, True) if premasked_out else y
    y = mods[8](mods[7](x))
    return (y, False) if premasked_out else y


def _conv_then_bn(conv, bn, x):
    """relu(bn(conv(x))) for a Conv2d followed by a training-mode BatchNorm2d: the conv's bias is a per-channel constant
    in front of a mean subtraction, so it is handed to the BatchNorm kernel as `input_shift` (running mean only) instead
    of being added to the activation and having a full-tensor reduction compute its (exactly zero) gradient."""
    fused_bn = (bn.training and x.is_cuda and conv.bias is not None and bn.affine and bn.track_running_stats
                and bn.momentum is not None and type(bn) is torch.nn.BatchNorm2d and type(conv) is torch.nn.Conv2d
                and conv.padding_mode == "zeros")
    if not fused_bn:
        return batchnorm_relu(bn, conv(x), True)
    y = _conv2d(conv, x)
    if not _nhwc_ok(y):
        return batchnorm_relu(bn, y + conv.bias.view(1, -1, 1, 1).to(y.dtype), True)
    return batchnorm_relu(bn, y, True, shift=conv.bias)
