"""ms_conv3x3_nhwc_bf16 vs MIOpen (torch conv2d, channels_last bf16) at the conv-branch shapes of MedMamba-T bs 64: correctness and
cold-cache time (a 1 GiB fill between calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from medical_image_classification_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.lib()
flush = torch.empty(1 << 28, device=dev, dtype=torch.float32)
def cold(fn, n=6):
    ts = []
    for _ in range(n):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = [(64, 128), (128, 64), (256, 32), (512, 16)] if len(sys.argv) > 2 and sys.argv[2] == "B" else [(48, 56), (96, 28), (192, 14), (384, 7)]
if len(sys.argv) > 3:
    shapes = [shapes[int(sys.argv[3])]]
for C, Hh in shapes:
    x = torch.randn(bs, C, Hh, Hh, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x)
    st = _lib.current_stream_ptr(dev)
    run = lambda: _lib.check(lib.ms_conv3x3_nhwc_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), bs, Hh, Hh, C, C, st), "conv")
    run(); torch.cuda.synchronize()
    ref = F.conv2d(x.float(), w.float(), padding=1)
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    t_ms = cold(run); t_mi = cold(lambda: F.conv2d(x, w, padding=1))
    print(f"C={C:4d} {Hh}x{Hh}: ms {t_ms:6.1f} us  MIOpen {t_mi:6.1f} us   max rel err {err:.2e}", flush=True)
    # weight gradient: ms_conv3x3_wgrad (main + finalize) vs MIOpen's (incl. its helper launches)
    if True:
        dy = torch.randn_like(x)
        ns = lib.ms_conv3x3_wgrad_scratch_floats(bs, Hh, Hh, C, C)
        scratch = torch.empty(ns, device=dev); dw = torch.empty(C, C, 3, 3, device=dev)
        runw = lambda: _lib.check(lib.ms_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), scratch.data_ptr(), ns, bs, Hh, Hh, C, C, st), "wgrad")
        runw(); torch.cuda.synchronize()
        refw = torch.ops.aten.convolution_backward(dy.float(), x.float(), w.float(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        errw = float((dw - refw).abs().max() / refw.abs().max())
        miw = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
        print(f"        wgrad: ms {cold(runw):6.1f} us  MIOpen {cold(miw):6.1f} us   max rel err {errw:.2e}", flush=True)
