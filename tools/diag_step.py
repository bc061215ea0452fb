"""Per-phase timing of one MedMamba-T training step on the GPU (diagnostic, prints as it goes)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from medical_image_classification_amd.train import build_model, synthetic_batch, train_step
def log(m): print(f"[{time.perf_counter()-T0:7.2f}s] {m}", flush=True)
T0 = time.perf_counter()
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ac = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else None
dev = torch.device("cuda:0")
net = build_model(num_classes=8).to(dev).train()
log("model built")
opt = torch.optim.Adam(net.parameters(), lr=1e-4); lossf = nn.CrossEntropyLoss()
x, y = synthetic_batch(bs, 8, 224, dev)
for i in range(3):
    t = time.perf_counter()
    if ac is not None:
        with torch.autocast("cuda", dtype=ac):
            out = net(x); loss = lossf(out, y)
    else:
        out = net(x); loss = lossf(out, y)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.step(); opt.zero_grad(set_to_none=True); torch.cuda.synchronize(); t3 = time.perf_counter()
    log(f"step {i}: fwd {1e3*(t1-t):.1f} ms  bwd {1e3*(t2-t1):.1f} ms  opt {1e3*(t3-t2):.1f} ms  loss {loss.item():.4f}")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    train_step(net, opt, lossf, x, y, ac); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70), flush=True)
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=70), flush=True)
# pure host cost of one step: enqueue without waiting for the GPU in between
torch.cuda.synchronize()
ts = []
for i in range(5):
    t = time.perf_counter(); train_step(net, opt, lossf, x, y, ac); ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
log("host enqueue time per step (GPU idle at start): " + " ".join(f"{1e3 * v:.1f}" for v in ts) + " ms")
