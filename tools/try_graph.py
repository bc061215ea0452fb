"""Does a full training step (fwd + bwd + Adam) capture into one HIP graph and replay correctly?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from medical_image_classification_amd.train import build_model, synthetic_batch
def log(m): print(f"[{time.perf_counter()-T0:7.2f}s] {m}", flush=True)
T0 = time.perf_counter()
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = build_model(num_classes=8).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True)
lossf = nn.CrossEntropyLoss()
x, y = synthetic_batch(bs, 8, 224, dev)

def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = lossf(net(x), y)
    loss.backward()
    opt.step()
    return loss

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3):
        l = step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize(); log(f"eager warmup done, loss {l.item():.4f}")
t = time.perf_counter()
for i in range(5): l = step()
torch.cuda.synchronize(); log(f"eager: {(time.perf_counter()-t)/5*1e3:.2f} ms/step, loss {l.item():.4f}")
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize(); log("captured")
for i in range(3): g.replay()
torch.cuda.synchronize(); log(f"replayed, loss {static_loss.item():.4f}")
t = time.perf_counter()
for i in range(10): g.replay()
torch.cuda.synchronize(); log(f"graph: {(time.perf_counter()-t)/10*1e3:.2f} ms/step, loss {static_loss.item():.4f}")
