"""Time ms_dwconv3x3_silu_nhwc_fwd/bwd at the MedMamba-T stage shapes (bs 64): the one-node SS2D call (4 du slabs + the x_proj
term in, bf16 dx out into the xz gradient) and the plain call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd import _lib

d = torch.device("cuda:0")
lib = _lib.lib()
st = _lib.current_stream_ptr(d)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, C) in ((64, 56, 96), (64, 28, 192), (64, 14, 384), (64, 7, 768)):
    W = H
    xz = torch.randn(B, H, W, 2 * C, device=d).bfloat16()
    w = torch.randn(C, 9, device=d); b = torch.randn(C, device=d)
    y = torch.empty(B, H, W, C, device=d)
    g4 = torch.randn(4, B, H, W, C, device=d); ge = torch.randn(B, H, W, C, device=d)
    dxz = torch.empty_like(xz); scratch = torch.empty(lib.ms_dwconv3x3_silu_nhwc_bwd_scratch_floats(B, C, H, W), device=d); dw = torch.zeros(C, 9, device=d); db = torch.zeros(C, device=d)
    dx32 = torch.empty_like(y)
    tf = timeit(lambda: lib.ms_dwconv3x3_silu_nhwc_fwd(xz.data_ptr(), 1, w.data_ptr(), b.data_ptr(), y.data_ptr(), B, C, H, W, 2 * C, st))
    t4 = timeit(lambda: lib.ms_dwconv3x3_silu_nhwc_bwd(xz.data_ptr(), 1, w.data_ptr(), b.data_ptr(), g4.data_ptr(), 4, B * H * W * C,
                                                       ge.data_ptr(), dxz.data_ptr(), 1, 2 * C, scratch.data_ptr(), dw.data_ptr(),
                                                       db.data_ptr(), B, C, H, W, 2 * C, st))
    t1 = timeit(lambda: lib.ms_dwconv3x3_silu_nhwc_bwd(xz.data_ptr(), 1, w.data_ptr(), b.data_ptr(), g4.data_ptr(), 1, 0,
                                                       None, dx32.data_ptr(), 0, C, scratch.data_ptr(), dw.data_ptr(),
                                                       db.data_ptr(), B, C, H, W, 2 * C, st))
    slab = B * H * W * C * 4 / 1e6
    print(f"B{B} H{H} C{C}: slab {slab:.1f} MB  fwd {tf:.1f} us  bwd(4 slabs + extra, bf16 dx) {t4:.1f} us  bwd(1 slab, fp32 dx) {t1:.1f} us", flush=True)
