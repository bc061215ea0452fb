"""cProfile of the host side of one training step (small batch: the GPU is never the bottleneck)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from medical_image_classification_amd.train import build_model, synthetic_batch, train_step
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net = build_model(num_classes=8).to(dev).train()
from medical_image_classification_amd.train import make_adam
opt = make_adam(net.parameters(), lr=1e-4); lossf = nn.CrossEntropyLoss()
x, y = synthetic_batch(bs, 8, 224, dev)
for _ in range(5): train_step(net, opt, lossf, x, y, torch.bfloat16)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): train_step(net, opt, lossf, x, y, torch.bfloat16)
torch.cuda.synchronize()
print(f"bs={bs}: {1e3 * (time.perf_counter() - t0) / 10:.2f} ms per step (host-bound)")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): train_step(net, opt, lossf, x, y, torch.bfloat16)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
