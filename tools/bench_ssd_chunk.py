"""Time the SSD operator forward (and backward when built) on csrc/ssd_chunk.hip against the torch chunked formulation and the
scan-kernel path, at the shapes of CNN_Mamba.VSSM (n = 64) and VFEFM (n = 512), batch 32.  usage: python tools/bench_ssd_chunk.py [bwd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd import cnn_mamba as cm
dev = torch.device("cuda:0")
bwd = len(sys.argv) > 1
def t(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for (b, l, h, n) in [(32, 3136, 8, 64), (32, 784, 16, 64), (32, 196, 32, 64), (32, 49, 64, 64), (32, 3136, 8, 512), (32, 784, 16, 512), (32, 196, 32, 512), (32, 49, 64, 512)]:
    p = 64
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(b, l, h, p, device=dev, generator=g, requires_grad=bwd)
    dt = (torch.randn(b, l, h, device=dev, generator=g) - 1.0).requires_grad_(bwd)
    A = (-(torch.rand(h, device=dev, generator=g) * 4 + 0.2)).requires_grad_(bwd)
    B = (torch.randn(b, l, 1, n, device=dev, generator=g) * 0.3).requires_grad_(bwd)
    C = (torch.randn(b, l, 1, n, device=dev, generator=g) * 0.3).requires_grad_(bwd)
    D = torch.randn(h, device=dev, generator=g, requires_grad=bwd)
    bias = (torch.randn(h, device=dev, generator=g) * 0.5).requires_grad_(bwd)
    gy = torch.randn(b, l, h, p, device=dev, generator=g)
    def run(fn):
        y = fn()
        if bwd:
            torch.autograd.grad(y, (x, dt, A, B, C, D, bias), gy)
    tk = t(lambda: run(lambda: cm._SSDChunkKernels.apply(x, dt, A, B, C, D, bias, True)))
    try:
        tt = t(lambda: run(lambda: cm._ssd_chunked(x, dt, A, B, C, D, bias, True)), it=3)
    except torch.OutOfMemoryError:
        tt = float("nan")
    print(f"b{b} l{l} h{h} n{n}: kernels {tk:8.2f} ms   torch chunked {tt:8.2f} ms ({'fwd+bwd' if bwd else 'fwd'})", flush=True)
