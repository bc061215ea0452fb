"""Do a side stream and the current stream execute concurrently?  Two 300-us single-workgroup spin kernels (ms_spin)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_classification_amd import _lib
dev = torch.device("cuda:0"); h = _lib.lib()
raw = lambda s: ctypes.c_void_p(s.cuda_stream)
cyc = 700_000        # ~300 us at 2.4 GHz
cur = torch.cuda.current_stream(dev)
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0)
solo = min(timed(lambda: h.ms_spin(cyc, raw(cur))) for _ in range(3))
print(f"solo {solo:.0f} us")
for i in range(8):
    s = torch.cuda.Stream(device=dev)
    both = min(timed(lambda: (h.ms_spin(cyc, raw(cur)), h.ms_spin(cyc, raw(s)))) for _ in range(3))
    print(f"stream {i} ({s.cuda_stream:#x}): both {both:.0f} us -> {'CONCURRENT' if both < 1.5 * solo else 'serialised'}", flush=True)
