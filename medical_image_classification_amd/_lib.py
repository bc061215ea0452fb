"""ctypes binding of libmedscan.so (the C ABI declared in include/medscan.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If it is missing, cannot be
loaded, or a call returns a non-zero status, a RuntimeError is raised -- loudly, never silently.
PyTorch is used only for device memory and streams: every call passes raw `data_ptr()`s plus the
current HIP stream, so the kernels are ordered with the surrounding torch ops and are graph-capturable.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("MEDSCAN_LIBRARY") or os.path.join(_HERE, "libmedscan.so")    # override: kernel-variant experiments
_CSRC = os.path.join(_HERE, "csrc")

c_i32, c_i64, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p


class MsScanParams(ctypes.Structure):
    """Mirror of `MsScanParams` (include/medscan.h); field set of the reference's SSMParamsBase
    (selective_scan.h:26-69)."""
    _fields_ = (
        [(n, c_i32) for n in ("batch", "dim", "seqlen", "dstate", "n_groups", "delta_softplus", "map_h", "map_w")]
        + [(n, c_i64) for n in (
            "u_batch_stride", "u_group_stride", "u_d_stride", "u_l_stride",
            "delta_batch_stride", "delta_group_stride", "delta_d_stride", "delta_l_stride",
            "out_batch_stride", "out_group_stride", "out_d_stride", "out_l_stride",
            "A_d_stride", "A_dstate_stride",
            "B_batch_stride", "B_group_stride", "B_dstate_stride", "B_l_stride",
            "C_batch_stride", "C_group_stride", "C_dstate_stride", "C_l_stride")]
        + [(n, c_vp) for n in ("u", "delta", "A", "B", "C", "D", "delta_bias", "out", "x", "dt_x", "dt_w")]
        + [(n, c_i32) for n in ("dt_rank", "segments")]
    )


class MsScanBwdParams(ctypes.Structure):
    """Mirror of `MsScanBwdParams`; field set of SSMParamsBwd (selective_scan.h:71-101)."""
    _fields_ = (
        [("f", MsScanParams)]
        + [(n, c_i64) for n in (
            "dout_batch_stride", "dout_group_stride", "dout_d_stride", "dout_l_stride",
            "du_batch_stride", "du_group_stride", "du_d_stride", "du_l_stride",
            "ddelta_batch_stride", "ddelta_group_stride", "ddelta_d_stride", "ddelta_l_stride",
            "dB_batch_stride", "dB_group_stride", "dB_dstate_stride", "dB_l_stride",
            "dC_batch_stride", "dC_group_stride", "dC_dstate_stride", "dC_l_stride")]
        + [(n, c_vp) for n in ("dout", "du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "ddt_x", "ddt_w")]
    )


class MsCastDesc(ctypes.Structure):
    """Mirror of `MsCastDesc` (one fp32 -> bf16 working copy of ms_cast_bf16_multi)."""
    _fields_ = [("src", c_vp), ("dst", c_vp), ("n", c_i64), ("inner", c_i32), ("taps", c_i32)]


class MsAdamDesc(ctypes.Structure):
    """Mirror of `MsAdamDesc` (one parameter tensor of ms_adam_multi)."""
    _fields_ = [("p", c_vp), ("m", c_vp), ("v", c_vp), ("n", c_i64)]


class MsBnFold(ctypes.Structure):
    """Mirror of `MsBnFold` (one BatchNorm folded into the convolutions around it)."""
    _fields_ = [("sums", c_vp), ("gamma", c_vp), ("beta", c_vp), ("shift", c_vp), ("running_mean", c_vp), ("running_var", c_vp),
                ("num_batches_tracked", c_vp), ("save_mean", c_vp), ("save_rstd", c_vp), ("momentum", ctypes.c_float), ("eps", ctypes.c_float)]


class MsBnBwd(ctypes.Structure):
    """Mirror of `MsBnBwd` (a BatchNorm's backward reduce riding on the convolution's input-gradient launch)."""
    _fields_ = [("x_pre", c_vp), ("x_pre_is_f32", ctypes.c_int), ("x_pre_pixel_stride", c_i64), ("gamma", c_vp), ("beta", c_vp),
                ("save_mean", c_vp), ("save_rstd", c_vp), ("relu", ctypes.c_int), ("sums", c_vp)]


BN_REPLICAS = 16
ADAM_CHUNK, ADAM_MAX_TENSORS = 4096, 448
CAST_CHUNK, CAST_TILE_O, CAST_TILE_I, CAST_TILE_MAX_TAPS = 2048, 64, 16, 9

EXPORTS = ("ms_selective_scan_fwd", "ms_selective_scan_bwd", "ms_scan_n_chunks", "ms_scan_seg_floats", "ms_cross_scan",
           "ms_cross_merge", "ms_cross_scan_nhwc", "ms_cross_merge_nhwc", "ms_dwconv3x3_silu_fwd", "ms_dwconv3x3_silu_bwd", "ms_dwconv3x3_silu_nhwc_fwd",
           "ms_dwconv3x3_silu_nhwc_bwd", "ms_dwconv3x3_silu_nhwc_bwd_scratch_floats", "ms_ln_gate_fwd", "ms_ln_gate_fwd_keep", "ms_ln_gate_bwd", "ms_layernorm_fwd", "ms_layernorm_bwd", "ms_layernorm_taps_fwd", "ms_layernorm_taps_bwd",
           "ms_block_tail_fwd", "ms_block_tail_bwd", "ms_dtproj_fwd", "ms_dtproj_fwd_act", "ms_dtproj_bwd", "ms_dtproj_bwd_scratch_floats", "ms_bn_relu_nhwc_fwd",
           "ms_bn_relu_nhwc_bwd", "ms_bn_scratch_floats", "ms_ssd_chunk_carry", "ms_ssd_chunk_fwd", "ms_ssd_chunk_fwd_off", "ms_ssd_chunk_bwd_off", "ms_ssd_chunk_bwd", "ms_rms_gate_fwd", "ms_rms_gate_bwd", "ms_gemm_bf16", "ms_gemm_f32", "ms_gemm_bf16_bias_act", "ms_gemm_bf16_wgrad_bias",
           "ms_cast_bf16_multi", "ms_patchify4_bf16", "ms_adam_multi", "ms_block_tail_bwd_relu", "ms_block_head_bwd", "ms_debug_gemm_tile", "ms_conv3x3_nhwc_bf16", "ms_conv3x3_wgrad", "ms_conv3x3_wgrad_scratch_floats", "ms_conv3x3_bn_nhwc_bf16", "ms_bn_apply_sums_nhwc", "ms_conv3x3_bnbwd_nhwc_bf16", "ms_bn_bwd_apply_sums_nhwc", "ms_gemm_bf16_bnbwd", "ms_linear_bwd_ok", "ms_linear_bwd_bf16", "ms_spin", "ms_abi_version", "ms_status_string")
ABI_VERSION = 9

_lib = None


def build(verbose=False):
    """Compile libmedscan.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", _CSRC, "-j4"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return _SO


def lib():
    """Load (once) and return the ctypes handle.  Raises RuntimeError if the library is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise RuntimeError(
            f"{_SO} is missing: the MI355X selective-scan kernels are not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            "medical_image_classification_amd/csrc`). There is no CPU fallback in the product path.")
    try:
        h = ctypes.CDLL(_SO)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"cannot load {_SO}: {e}") from e
    for name in EXPORTS:
        if not hasattr(h, name):
            raise RuntimeError(f"{_SO} does not export {name}")
    h.ms_selective_scan_fwd.argtypes = [ctypes.POINTER(MsScanParams), c_vp]
    h.ms_selective_scan_bwd.argtypes = [ctypes.POINTER(MsScanBwdParams), c_vp]
    h.ms_scan_n_chunks.argtypes = [ctypes.c_int]
    h.ms_cross_scan.argtypes = [c_vp, c_vp] + [ctypes.c_int] * 4 + [c_vp]
    h.ms_cross_merge.argtypes = [c_vp, c_vp] + [ctypes.c_int] * 4 + [c_vp]
    h.ms_cross_scan_nhwc.argtypes = [c_vp, c_i64, c_vp] + [ctypes.c_int] * 4 + [c_vp]
    h.ms_cross_merge_nhwc.argtypes = [c_vp, c_vp, c_i64] + [ctypes.c_int] * 4 + [c_vp]
    h.ms_dwconv3x3_silu_fwd.argtypes = [c_vp] * 4 + [ctypes.c_int] * 4 + [c_vp]
    h.ms_dwconv3x3_silu_bwd.argtypes = [c_vp] * 7 + [ctypes.c_int] * 4 + [c_vp]
    h.ms_dwconv3x3_silu_nhwc_fwd.argtypes = [c_vp, ctypes.c_int] + [c_vp] * 3 + [ctypes.c_int] * 4 + [c_i64, c_vp]
    h.ms_dwconv3x3_silu_nhwc_bwd.argtypes = ([c_vp, ctypes.c_int, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_vp, ctypes.c_int,
                                              c_i64, c_vp, c_vp, c_vp] + [ctypes.c_int] * 4 + [c_i64, c_vp])
    h.ms_dwconv3x3_silu_nhwc_bwd_scratch_floats.argtypes = [ctypes.c_int] * 4
    c_f, c_int = ctypes.c_float, ctypes.c_int
    h.ms_ln_gate_fwd.argtypes = [c_vp, c_i64, c_vp, c_int, c_i64, c_vp, c_vp, c_f, c_vp, c_int, c_i64, c_int, c_vp]
    h.ms_ln_gate_fwd_keep.argtypes = [c_vp, c_i64, c_vp, c_int, c_i64, c_vp, c_vp, c_f, c_vp, c_int, c_vp, c_i64, c_int, c_vp]
    h.ms_ln_gate_bwd.argtypes = [c_vp, c_i64, c_vp, c_int, c_i64, c_vp, c_vp, c_f, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_vp,
                                 c_i64, c_int, c_vp]
    h.ms_layernorm_fwd.argtypes = [c_vp, c_i64, c_vp, c_vp, c_f, c_vp, c_int, c_i64, c_int, c_vp]
    h.ms_layernorm_taps_fwd.argtypes = [c_vp, c_vp, c_vp, c_f, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]
    h.ms_layernorm_taps_bwd.argtypes = [c_vp, c_vp, c_f, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]
    h.ms_layernorm_bwd.argtypes = [c_vp, c_i64, c_vp, c_f, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp]
    h.ms_block_tail_fwd.argtypes = [c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_vp]
    h.ms_block_tail_bwd.argtypes = [c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_i64, c_i64, c_int, c_vp]
    h.ms_dtproj_fwd.argtypes = [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp]
    h.ms_dtproj_fwd_act.argtypes = [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp]
    h.ms_dtproj_bwd.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp]
    h.ms_dtproj_bwd_scratch_floats.argtypes = [c_i64, c_int, c_int]
    h.ms_bn_relu_nhwc_fwd.argtypes = [c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f, c_f, c_int, c_vp, c_int, c_vp, c_vp, c_vp,
                                      c_i64, c_int, c_vp]
    h.ms_bn_relu_nhwc_bwd.argtypes = [c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp]
    h.ms_block_tail_bwd_relu.argtypes = [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_i64, c_i64, c_int, c_vp]
    h.ms_block_head_bwd.argtypes = [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_i64, c_int, c_vp]
    h.ms_cast_bf16_multi.argtypes = [c_vp, c_vp, c_int, c_vp]
    h.ms_gemm_bf16_bias_act.argtypes = [c_vp, c_int, c_int, c_i64, c_vp, c_int, c_int, c_i64, c_vp, c_int, c_i64, c_int, c_int, c_int, c_vp, c_int, c_vp]
    h.ms_bn_scratch_floats.argtypes = [c_int]
    h.ms_rms_gate_fwd.argtypes = [c_vp, c_i64, c_int, c_vp, c_int, c_i64, c_vp, c_f, c_vp, c_int, c_i64, c_int, c_vp]
    h.ms_rms_gate_bwd.argtypes = [c_vp, c_i64, c_int, c_vp, c_int, c_i64, c_vp, c_f, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_vp]
    h.ms_ssd_chunk_carry.argtypes = [c_vp] * 5 + [c_int] * 7 + [c_vp]
    h.ms_gemm_bf16_wgrad_bias.argtypes = [c_vp, c_int, c_i64, c_vp, c_int, c_i64, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, c_vp]
    h.ms_gemm_bf16.argtypes = [c_vp, c_int, c_int, c_i64, c_vp, c_int, c_int, c_i64, c_vp, c_int, c_i64, c_int, c_int, c_int, c_int, c_vp]
    h.ms_ssd_chunk_fwd.argtypes = [c_vp] * 6 + [c_int] + [c_vp] * 6 + [c_int] * 5 + [c_vp]
    h.ms_ssd_chunk_fwd_off.argtypes = [c_vp] * 5 + [c_int, c_vp] + [c_int] * 5 + [c_vp]
    h.ms_ssd_chunk_bwd_off.argtypes = [c_vp] * 6 + [c_int] * 5 + [c_vp]
    h.ms_ssd_chunk_bwd.argtypes = [c_vp] * 13 + [c_int, c_int] + [c_vp] * 9 + [c_int] * 5 + [c_vp]
    h.ms_gemm_f32.argtypes = [c_vp, c_int, c_i64, c_vp, c_int, c_i64, c_vp, c_int, c_i64, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp]
    h.ms_patchify4_bf16.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]
    h.ms_adam_multi.argtypes = [c_vp, c_vp, c_int, c_vp, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_vp]
    h.ms_debug_gemm_tile.argtypes = [c_int, c_int]
    h.ms_conv3x3_nhwc_bf16.argtypes = [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]
    h.ms_conv3x3_wgrad.argtypes = [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_int, c_vp]
    h.ms_conv3x3_wgrad_scratch_floats.argtypes = [c_int, c_int, c_int, c_int, c_int]
    h.ms_conv3x3_bn_nhwc_bf16.argtypes = [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]
    h.ms_bn_apply_sums_nhwc.argtypes = [c_vp, c_vp, c_int, c_vp, c_i64, c_int, c_vp]
    h.ms_conv3x3_bnbwd_nhwc_bf16.argtypes = [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]
    h.ms_bn_bwd_apply_sums_nhwc.argtypes = [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_int, c_vp]
    h.ms_gemm_bf16_bnbwd.argtypes = [c_vp, c_int, c_i64, c_vp, c_int, c_i64, c_vp, c_int, c_i64, c_int, c_int, c_int, c_vp, c_vp]
    h.ms_scan_seg_floats.argtypes = [c_int, c_int, c_int]
    h.ms_linear_bwd_ok.argtypes = [c_int, c_int]
    h.ms_linear_bwd_bf16.argtypes = [c_vp, c_int, c_i64, c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_int, c_int, c_int, c_vp]
    h.ms_spin.argtypes = [ctypes.c_longlong, c_vp]
    h.ms_status_string.restype = ctypes.c_char_p
    h.ms_status_string.argtypes = [ctypes.c_int]
    for name in EXPORTS[:-1]:
        getattr(h, name).restype = ctypes.c_int
    h.ms_dwconv3x3_silu_nhwc_bwd_scratch_floats.restype = c_i64
    h.ms_dtproj_bwd_scratch_floats.restype = c_i64
    h.ms_scan_seg_floats.restype = c_i64
    h.ms_conv3x3_wgrad_scratch_floats.restype = c_i64
    if h.ms_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{_SO}: ABI version {h.ms_abi_version()} != {ABI_VERSION} (stale build?)")
    _lib = h
    return h


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed: {lib().ms_status_string(status).decode()} (status {status})")


def current_stream_ptr(device):
    """HIP stream the kernels are enqueued on: torch's current stream of `device` (raw handle, no Stream object)."""
    import torch
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device()))


class on_device:
    """`with on_device(dev):` -- torch.cuda.device(dev) only when `dev` is not already the current device (the common
    single-GPU-per-process case costs one integer compare instead of two driver calls)."""
    __slots__ = ("ctx",)

    def __init__(self, device):
        import torch
        idx = device.index
        self.ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "medical_image_classification_amd: this operator runs only on an MI355X (HIP) device; got a "
                f"{t.device} tensor. There is no CPU fallback in the product path (the CPU oracle lives in "
                "oracle/ and is test infrastructure).")
