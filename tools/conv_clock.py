"""In-kernel timeline of conv3x3_nhwc_kernel (diagnostic build: tools/build_variant.sh convclk "conv3x3.hip" "-DMS_CONV_CLOCK=1";
MEDSCAN_LIBRARY=build/variants/libmedscan_convclk.so python3 tools/conv_clock.py [stage]): s_memtime stamps of thread 0 of every workgroup."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from medical_image_classification_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.lib()
raw = ctypes.CDLL(os.environ["MEDSCAN_LIBRARY"])
bs = 64
for si in ([int(sys.argv[1])] if len(sys.argv) > 1 else [0, 1, 2, 3]):
    C, Hh = [(48, 56), (96, 28), (192, 14), (384, 7)][si]
    x = torch.randn(bs, C, Hh, Hh, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x); st = _lib.current_stream_ptr(dev)
    flush = torch.empty(1 << 28, device=dev, dtype=torch.float32)
    for _ in range(3):
        flush.fill_(1.0)
        _lib.check(lib.ms_conv3x3_nhwc_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), bs, Hh, Hh, C, C, st), "conv")
        torch.cuda.synchronize()
    ntile = bs * ((Hh + 15) // 16) * ((Hh + 7) // 8)
    n = min(ntile, 8192)
    buf = np.zeros(n * 32, dtype=np.int64)
    assert raw.ms_debug_conv_clock(buf.ctypes.data_as(ctypes.c_void_p), n * 32) == 0
    t = buf.reshape(n, 32).astype(np.float64)
    ns = (C + 31) // 32
    t0 = t[:, 0].min()
    print(f"stage {si}: C={C} {Hh}x{Hh}, {ntile} tiles, {ns} slices; memtime ticks (100 MHz => x10 ns), medians over workgroups")
    print(f"  first workgroup starts 0, last starts {t[:, 0].max() - t0:.0f}, last ends {t[:, 31].max() - t0:.0f}")
    life = t[:, 31] - t[:, 0]
    print(f"  workgroup lifetime: median {np.median(life):.0f}  min {life.min():.0f}  max {life.max():.0f}")
    print(f"  entry -> offsets + first fetch issued: {np.median(t[:, 1] - t[:, 0]):.0f}; -> first barrier passed: {np.median(t[:, 2] - t[:, 0]):.0f}")
    for k in range(ns):
        b = 2 + 4 * k
        print(f"  slice {k}: put (wait loads + LDS writes) {np.median(t[:, b + 1] - t[:, b]):.0f}  barrier {np.median(t[:, b + 2] - t[:, b + 1]):.0f}  fetch + products {np.median(t[:, b + 3] - t[:, b + 2]):.0f}" +
              (f"  barrier before next {np.median(t[:, b + 4] - t[:, b + 3]):.0f}" if k + 1 < ns else f"  stores issued {np.median(t[:, 31] - t[:, b + 3]):.0f}"))
