"""Which Python line issues the small aten ops (casts, fills, copies) of one training step: a TorchDispatchMode logs every
aten call with the innermost package frame (ops issued by C++ autograd nodes have no Python frame: "(autograd engine)").
  python tools/launch_audit.py [batch]        -> table sorted by calls"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from torch.utils._python_dispatch import TorchDispatchMode
from medical_image_classification_amd.train import build_model, synthetic_batch, train_step, make_adam

PKG = "medical_image_classification_amd"
WATCH = ("copy_", "_to_copy", "fill_", "zero_", "zeros", "zeros_like", "sum", "add_", "add", "clone", "contiguous", "mul", "div",
         "empty_like", "cat", "clamp_min", "threshold_backward", "bernoulli_", "div_", "mm", "bmm", "addmm", "new_zeros", "full")


class Audit(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.rows = collections.Counter(); self.shapes = {}
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            site = "(autograd engine)"
            for fr in reversed(traceback.extract_stack()):
                if PKG in fr.filename and "launch_audit" not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"; break
            t = next((a for a in args if isinstance(a, torch.Tensor)), None)
            desc = "" if t is None else f"{tuple(t.shape)} {str(t.dtype)[6:]}{'' if t.is_contiguous() else ' nc' + str(tuple(t.stride()))}"
            if name in ("_to_copy", "copy_") and t is not None:
                dst = (kwargs or {}).get("dtype", None)
                src = args[1] if name == "copy_" and len(args) > 1 and isinstance(args[1], torch.Tensor) else None
                desc += f" -> {str(dst)[6:] if dst else ''}{'' if src is None else ' <- ' + str(src.dtype)[6:] + (' nc' if not src.is_contiguous() else '')}"
            self.rows[(name, site)] += 1; self.shapes.setdefault((name, site), collections.Counter())[desc] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net = build_model(num_classes=8).to(dev).train()
opt = make_adam(net.parameters(), lr=1e-4); lossf = nn.CrossEntropyLoss()
x, y = synthetic_batch(bs, 8, 224, dev)
for _ in range(3): train_step(net, opt, lossf, x, y, torch.bfloat16)
torch.cuda.synchronize()
a = Audit()
with a:
    train_step(net, opt, lossf, x, y, torch.bfloat16)
torch.cuda.synchronize()
tot = collections.Counter()
for (name, site), n in a.rows.items(): tot[name] += n
print("per op:", ", ".join(f"{k} {v}" for k, v in tot.most_common()))
for (name, site), n in sorted(a.rows.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:18s} {site}")
    for desc, k in a.shapes[(name, site)].most_common(12):
        print(f"          {k:3d} x {desc}")
