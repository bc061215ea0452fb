"""CPU-only: libmedscan.so builds/loads and exports every symbol include/medscan.h declares; the ctypes
mirrors of the parameter structs have the C layout; host-side operand validation raises loudly."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from medical_image_classification_amd import _lib
    if not os.path.exists(_lib._SO):
        _lib.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "medscan.h")).read()
    declared = set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", hdr))
    assert {"ms_selective_scan_fwd", "ms_selective_scan_bwd", "ms_cross_scan", "ms_cross_merge",
            "ms_dwconv3x3_silu_fwd", "ms_dwconv3x3_silu_bwd"} <= declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in medscan.h but not exported"
    assert lib.ms_abi_version() == 9
    assert lib.ms_scan_n_chunks(3136) == 98 and lib.ms_scan_n_chunks(49) == 2 and lib.ms_scan_n_chunks(0) == 0
    assert lib.ms_status_string(-3).decode() == "unsupported state dimension"


def test_struct_layout_matches_c(tmp_path):
    """sizeof/offsetof of the ctypes mirrors == what a C compiler sees in medscan.h."""
    from medical_image_classification_amd._lib import (ADAM_CHUNK, ADAM_MAX_TENSORS, BN_REPLICAS, MsAdamDesc, MsBnBwd, MsBnFold, MsCastDesc, MsScanBwdParams,
                                                       MsScanParams)
    src = tmp_path / "lay.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "medscan.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(MsScanParams),offsetof(MsScanParams,u),offsetof(MsScanParams,x),sizeof(MsScanBwdParams),'
                   'offsetof(MsScanBwdParams,dout),offsetof(MsScanBwdParams,ddelta_bias));'
                   'printf("%zu %zu %zu\\n",offsetof(MsScanParams,dt_w),offsetof(MsScanParams,dt_rank),offsetof(MsScanBwdParams,ddt_w));'
                   'printf("%zu %zu %zu\\n",sizeof(MsCastDesc),offsetof(MsCastDesc,n),offsetof(MsCastDesc,taps));'
                   'printf("%zu %zu %d %d\\n",sizeof(MsAdamDesc),offsetof(MsAdamDesc,n),MS_ADAM_CHUNK,MS_ADAM_MAX_TENSORS);'
                   'printf("%zu %zu %zu %d %d\\n",sizeof(MsBnFold),offsetof(MsBnFold,save_mean),offsetof(MsBnFold,eps),MS_BN_REPLICAS,MS_BN_FOLD_FLOATS(48));'
                   'printf("%zu %zu %zu %zu\\n",sizeof(MsBnBwd),offsetof(MsBnBwd,x_pre_pixel_stride),offsetof(MsBnBwd,relu),offsetof(MsBnBwd,sums));return 0;}\n')
    exe = tmp_path / "lay"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(MsScanParams), MsScanParams.u.offset, MsScanParams.x.offset,
            ctypes.sizeof(MsScanBwdParams), MsScanBwdParams.dout.offset, MsScanBwdParams.ddelta_bias.offset,
            MsScanParams.dt_w.offset, MsScanParams.dt_rank.offset, MsScanBwdParams.ddt_w.offset,
            ctypes.sizeof(MsCastDesc), MsCastDesc.n.offset, MsCastDesc.taps.offset,
            ctypes.sizeof(MsAdamDesc), MsAdamDesc.n.offset, ADAM_CHUNK, ADAM_MAX_TENSORS,
            ctypes.sizeof(MsBnFold), MsBnFold.save_mean.offset, MsBnFold.eps.offset, BN_REPLICAS, (2 * BN_REPLICAS + 1) * 48,
            ctypes.sizeof(MsBnBwd), MsBnBwd.x_pre_pixel_stride.offset, MsBnBwd.relu.offset, MsBnBwd.sums.offset]
    assert got == want


def test_null_and_shape_errors_without_gpu(lib):
    from medical_image_classification_amd._lib import MsScanParams
    assert lib.ms_selective_scan_fwd(None, None) == -1
    p = MsScanParams()
    assert lib.ms_selective_scan_fwd(ctypes.byref(p), None) == -1      # NULL operands
    assert lib.ms_cross_scan(None, None, 1, 1, 1, 1, None) == -1
    assert lib.ms_dwconv3x3_silu_fwd(None, None, None, None, 1, 1, 1, 1, None) == -1
    assert lib.ms_adam_multi(None, None, 1, None, 1, 1e-3, 1.0, 0.1, 0.999, 1e-3, 1e-8, None) == -1
    assert lib.ms_adam_multi(None, None, 1, None, 449, 1e-3, 1.0, 0.1, 0.999, 1e-3, 1e-8, None) == -2      # more tensors than one launch carries
    # the convolutions address with 32-bit byte offsets: sizes beyond that are refused before anything is launched (include/medscan.h)
    dummy = ctypes.c_void_p(64)
    assert lib.ms_conv3x3_nhwc_bf16(dummy, dummy, dummy, 1, 32768, 32768, 16, 16, None) == -6
    assert lib.ms_conv3x3_nhwc_bf16(dummy, dummy, dummy, 1, 8, 8, 24, 16, None) == -2                           # Ci not a multiple of 16
    assert lib.ms_conv3x3_wgrad(dummy, dummy, ctypes.cast(dummy, ctypes.POINTER(ctypes.c_float)), ctypes.cast(dummy, ctypes.POINTER(ctypes.c_float)),
                                1 << 40, 4096, 1024, 1024, 16, 16, None) == -6


def test_cpu_tensors_fail_loudly():
    from medical_image_classification_amd import selective_scan_fn
    u = torch.randn(1, 4, 8); A = -torch.rand(4, 2); B = torch.randn(1, 2, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        selective_scan_fn(u, u, A, B, B)
