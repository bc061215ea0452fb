"""Time ms_selective_scan_fwd / _bwd alone (HIP events around the bare C-ABI launches, SS2D addressing mode) at the
MedMamba-T stage shapes.  MEDSCAN_LIBRARY=<path> selects a kernel-variant build.
usage: python tools/scan_kernel_bench.py [batch] [iters] [stages e.g. 0,1,2,3] [variant: T|B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd import _lib
from medical_image_classification_amd._lib import MsScanBwdParams, MsScanParams
from medical_image_classification_amd.ss2d_fused import _ss2d_params
from medical_image_classification_amd.selective_scan_interface import algorithmic_bytes

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 3]
variant = sys.argv[4] if len(sys.argv) > 4 else "T"
stages = {"T": [(96, 56, 3, 2), (192, 28, 6, 2), (384, 14, 12, 4), (768, 7, 24, 2)],
          "B": [(128, 128, 4, 2), (256, 64, 8, 2), (512, 32, 16, 12), (1024, 16, 32, 2)]}[variant]
lib = _lib.lib()
tot = [0.0, 0.0, 0.0, 0.0]
for si in which:
    D, Hh, R, nblk = stages[si]
    L, N, C = Hh * Hh, 16, R + 32
    gen = torch.Generator(device=dev).manual_seed(0)
    A = torch.log(torch.arange(1, N + 1, device=dev, dtype=torch.float32)).repeat(4 * D, 1).contiguous()   # A_logs init
    Dp = torch.ones(4 * D, device=dev)
    bias = torch.rand(4 * D, device=dev, generator=gen) - 4.0
    xc = torch.randn(bs, Hh, Hh, D, device=dev, generator=gen)
    proj = torch.randn(bs, L, 4, C, device=dev, generator=gen)
    delta = torch.randn(4, bs, L, D, device=dev, generator=gen)
    gy = torch.randn(bs, L, D, device=dev, generator=gen)
    y4 = torch.empty(4, bs, L, D, device=dev)
    xs = torch.empty(bs, lib.ms_scan_n_chunks(L), N, 4 * D, device=dev)
    du4, dd4 = torch.empty_like(y4), torch.empty_like(y4)
    dproj = torch.zeros_like(proj)
    dA, dD, db = torch.zeros_like(A), torch.zeros_like(Dp), torch.zeros_like(bias)
    P = MsScanParams(); _ss2d_params(P, xc, proj, delta, A, Dp, bias, y4, xs, Hh, Hh, N, R, a_is_log=True)
    Q = MsScanBwdParams(); _ss2d_params(Q.f, xc, proj, delta, A, Dp, bias, None, xs, Hh, Hh, N, R, a_is_log=True)
    Q.dout_batch_stride, Q.dout_group_stride, Q.dout_d_stride, Q.dout_l_stride = L * D, 0, 1, D
    Q.du_batch_stride, Q.du_group_stride, Q.du_d_stride, Q.du_l_stride = L * D, bs * L * D, 1, D
    Q.ddelta_batch_stride, Q.ddelta_group_stride, Q.ddelta_d_stride, Q.ddelta_l_stride = L * D, bs * L * D, 1, D
    Q.dB_batch_stride, Q.dB_group_stride, Q.dB_dstate_stride, Q.dB_l_stride = L * 4 * C, C, 1, 4 * C
    Q.dC_batch_stride, Q.dC_group_stride, Q.dC_dstate_stride, Q.dC_l_stride = L * 4 * C, C, 1, 4 * C
    Q.dout, Q.du, Q.ddelta = gy.data_ptr(), du4.data_ptr(), dd4.data_ptr()
    Q.dA, Q.dD, Q.ddelta_bias = dA.data_ptr(), dD.data_ptr(), db.data_ptr()
    Q.dB, Q.dC = dproj.data_ptr() + 4 * R, dproj.data_ptr() + 4 * (R + N)
    if os.environ.get("KB_PRE", "1") == "1":     # what the training step launches: delta holds delta' = softplus(delta + bias) (MS_SCAN_DELTA_ACTIVATED)
        delta.copy_(torch.nn.functional.softplus(delta + bias.view(4, 1, 1, D)))
        P.delta_softplus |= 512; Q.f.delta_softplus |= 512
    st = _lib.current_stream_ptr(dev)
    def run(fn, arg):
        for _ in range(2):
            _lib.check(fn(ctypes.byref(arg), st), "scan")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn(ctypes.byref(arg), st)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    tf = run(lib.ms_selective_scan_fwd, P)
    tb = run(lib.ms_selective_scan_bwd, Q)
    if hasattr(lib, "ms_debug_clock"):           # diagnostic build (-DMS_CLOCK): the clock held during the forward kernel
        cbuf = (ctypes.c_ulonglong * 2)()
        torch.cuda.synchronize(); lib.ms_debug_clock(cbuf, 1)
        for _ in range(20):
            lib.ms_selective_scan_fwd(ctypes.byref(P), st)
        torch.cuda.synchronize(); lib.ms_debug_clock(cbuf, 1)
        print(f"   in-kernel clock of the forward scan: {cbuf[0] / max(1, cbuf[1]) * 0.1:.2f} GHz (s_memtime / s_memrealtime x 100 MHz)")
    if hasattr(lib, "ms_debug_stamps"):          # diagnostic build (-DMS_STAMP): cycles per wave per chunk, by phase
        buf = (ctypes.c_ulonglong * 8)()
        for name, fn, arg, rd in (("fwd", lib.ms_selective_scan_fwd, P, lib.ms_debug_stamps),
                                  ("bwd", lib.ms_selective_scan_bwd, Q, lib.ms_debug_stamps_bwd)):
            torch.cuda.synchronize(); rd(buf, 1)
            fn(ctypes.byref(arg), st); torch.cuda.synchronize(); rd(buf, 1)
            nch = lib.ms_scan_n_chunks(L)
            nw = bs * 4 * ((D + 7) // 8 if (name == "bwd" or bs * 4 * ((D + 15) // 16) < 2048) else (D + 15) // 16)
            print(f"   {name} stamps (cycles per wave per chunk, {nw} waves): " +
                  " ".join(f"[{i}] {buf[i] / nw / nch:7.0f}" for i in range(8)) + f"  sum {sum(buf) / nw / nch:7.0f}")
    bf, bb = algorithmic_bytes(bs, 4 * D, L, N, 4, False), algorithmic_bytes(bs, 4 * D, L, N, 4, True)
    print(f"stage {si} D={D:4d} L={L:5d}: fwd {tf*1e3:8.1f} us {bf/tf/1e6:7.0f} GB/s ({bf/tf/8e9*100:4.1f}%) | "
          f"bwd {tb*1e3:8.1f} us {bb/tb/1e6:7.0f} GB/s ({bb/tb/8e9*100:4.1f}%)", flush=True)
    tot[0] += nblk * tf; tot[1] += nblk * tb; tot[2] += nblk * bf; tot[3] += nblk * bb
if len(which) == 4:
    print(f"per step: fwd {tot[0]:.3f} ms ({tot[2]/tot[0]/8e9*100:.1f}% of 8 TB/s)  bwd {tot[1]:.3f} ms ({tot[3]/tot[1]/8e9*100:.1f}%)")
