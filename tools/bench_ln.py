"""Times ms_ln_gate_fwd/bwd and ms_layernorm_fwd/bwd at the MedMamba-T stage shapes (bs 64, bf16 I/O as under autocast)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd import _lib
from medical_image_classification_amd.block_ops import layernorm_rows

dev = torch.device("cuda:0")
h = _lib.lib()
st = lambda: _lib.current_stream_ptr(dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
SHAPES = [(96, 56, 64), (192, 28, 64), (384, 14, 64), (768, 7, 64)]            # MedMamba-T bs 64
if len(sys.argv) > 1 and sys.argv[1] == "B":                                      # MedMamba-B 512x512 bs 32
    SHAPES = [(128, 128, 32), (256, 64, 32), (512, 32, 32), (1024, 16, 32)]
for D, Hh, bs in SHAPES:
    npix = bs * Hh * Hh
    y4 = torch.randn(4, npix, D, device=dev); z = torch.randn(npix, D, device=dev).bfloat16()
    gm, bt = torch.randn(D, device=dev), torch.randn(D, device=dev)
    out = torch.empty(npix, D, device=dev, dtype=torch.bfloat16); dout = torch.randn(npix, D, device=dev).bfloat16()
    dy = torch.empty(npix, D, device=dev); dz = torch.empty_like(z); dgb = torch.zeros(2, D, device=dev)
    f = lambda: h.ms_ln_gate_fwd(y4.data_ptr(), npix * D, z.data_ptr(), 1, D, gm.data_ptr(), bt.data_ptr(), 1e-5, out.data_ptr(), 1, npix, D, st())
    b = lambda: h.ms_ln_gate_bwd(y4.data_ptr(), npix * D, z.data_ptr(), 1, D, gm.data_ptr(), bt.data_ptr(), 1e-5, dout.data_ptr(), 1,
                                 dy.data_ptr(), dz.data_ptr(), D, dgb[0].data_ptr(), dgb[1].data_ptr(), npix, D, st())
    bytes_b = npix * D * (16 + 2 + 2 + 4 + 2)
    tf, tb = timeit(f), timeit(b)
    ysum = torch.empty(npix, D, device=dev)
    fk = lambda: h.ms_ln_gate_fwd_keep(y4.data_ptr(), npix * D, z.data_ptr(), 1, D, gm.data_ptr(), bt.data_ptr(), 1e-5, out.data_ptr(), 1, ysum.data_ptr(), npix, D, st())
    bm = lambda: h.ms_ln_gate_bwd(ysum.data_ptr(), 0, z.data_ptr(), 1, D, gm.data_ptr(), bt.data_ptr(), 1e-5, dout.data_ptr(), 1,
                                  dy.data_ptr(), dz.data_ptr(), D, dgb[0].data_ptr(), dgb[1].data_ptr(), npix, D, st())
    tfk, tbm = timeit(fk), timeit(bm)
    Dh = D // 2
    x = torch.randn(npix, 2 * Dh, device=dev); g2, b2 = torch.randn(Dh, device=dev), torch.randn(Dh, device=dev)
    o2 = torch.empty(npix, Dh, device=dev, dtype=torch.bfloat16); do2 = torch.randn(npix, Dh, device=dev).bfloat16()
    dx2 = torch.empty(npix, Dh, device=dev)
    xr = x[:, Dh:]
    lf = lambda: h.ms_layernorm_fwd(xr.data_ptr(), 2 * Dh, g2.data_ptr(), b2.data_ptr(), 1e-6, o2.data_ptr(), 1, npix, Dh, st())
    lb = lambda: h.ms_layernorm_bwd(xr.data_ptr(), 2 * Dh, g2.data_ptr(), 1e-6, do2.data_ptr(), 1, dx2.data_ptr(), dgb[0].data_ptr(),
                                    dgb[1].data_ptr(), npix, Dh, st())
    tlf, tlb = timeit(lf), timeit(lb)
    print(f"D={D:4d} npix={npix:7d}: ln_gate fwd {tf:7.1f} (keep {tfk:6.1f}) us  bwd {tb:7.1f} us ({bytes_b / tb / 1e6:5.2f} TB/s; merged {tbm:6.1f} us = {npix * D * 14 / tbm / 1e6:5.2f} TB/s) | ln fwd {tlf:6.1f} us  bwd {tlb:6.1f} us "
          f"({npix * Dh * 10 / tlb / 1e6:5.2f} TB/s)", flush=True)
