"""Times ms_dtproj_fwd/bwd against the batched-GEMM formulation at the MedMamba-T stage shapes (bs 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd import _lib
dev = torch.device("cuda:0"); h = _lib.lib(); st = lambda: _lib.current_stream_ptr(dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for D, Hh, R in [(96, 56, 3), (192, 28, 6), (384, 14, 12), (768, 7, 24)]:
    npix, C = 64 * Hh * Hh, R + 32
    proj = torch.randn(npix, 4, C, device=dev); W = torch.randn(4, D, R, device=dev); dd = torch.randn(4, npix, D, device=dev)
    delta = torch.empty(4, npix, D, device=dev); dproj = torch.zeros_like(proj); dW = torch.zeros_like(W)
    tf = timeit(lambda: h.ms_dtproj_fwd(proj.data_ptr(), W.data_ptr(), delta.data_ptr(), npix, D, R, C, st()))
    ns = h.ms_dtproj_bwd_scratch_floats(npix, D, R); scr = torch.empty(max(ns, 1), device=dev)
    tb = timeit(lambda: h.ms_dtproj_bwd(dd.data_ptr(), proj.data_ptr(), W.data_ptr(), dproj.data_ptr(), dW.data_ptr(), scr.data_ptr(), ns, npix, D, R, C, st()))
    dts = proj[:, :, :R].permute(1, 0, 2).contiguous()
    S = next((c for c in (64, 32, 16, 8, 4, 2) if npix % c == 0 and npix // c >= 1024), 1)
    rf = timeit(lambda: torch.bmm(dts, W.transpose(1, 2)))
    rb = timeit(lambda: (torch.bmm(dd, W), torch.bmm(dd.view(4 * S, npix // S, D).transpose(1, 2), dts.view(4 * S, npix // S, R)).view(4, S, D, R).sum(1)))
    print(f"D={D:4d} R={R:2d} npix={npix:7d}: kernel fwd {tf:7.1f} bwd {tb:7.1f} us | bmm fwd {rf:7.1f} bwd {rb:7.1f} us", flush=True)
