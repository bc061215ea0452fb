"""Does a freshly created HIP stream run concurrently with the current stream?  (hardware-queue aliasing probe)"""
import time, torch
dev = torch.device("cuda:0")
a = torch.randn(256, device=dev); b = torch.randn(256, device=dev)
def chain(t, n=300):
    for _ in range(n): t.mul_(1.0001)
def measure(side):
    torch.cuda.synchronize(); t0 = time.perf_counter(); chain(a); torch.cuda.synchronize(); solo = time.perf_counter() - t0
    cur = torch.cuda.current_stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    side.wait_stream(cur)
    with torch.cuda.stream(side): chain(b)
    chain(a)
    cur.wait_stream(side); torch.cuda.synchronize(); both = time.perf_counter() - t0
    return solo, both
for i in range(6):
    s = torch.cuda.Stream(device=dev)
    for _ in range(2): r = measure(s)
    print(f"stream {i} id={s.cuda_stream:#x}: solo {1e3*r[0]:.2f} ms, both {1e3*r[1]:.2f} ms, ratio {r[1]/r[0]:.2f}", flush=True)
