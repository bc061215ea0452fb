#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by RUNNING the reference on CPU in this container.

The reference (/root/reference) never travels to the GPU box; only the small
input/output vectors written here do.  Nothing from the reference is copied:
this script imports its modules (with two stub modules for packages that are
absent in this image) and records what they compute on seeded inputs.

Import recipe:
  * `selective_scan_interface.py` hard-imports the CUDA extension
    (`selective_scan_interface.py:16`) -> register an empty `selective_scan_cuda`
    module and load the FILE, not the `mamba_ssm` package.
  * `MedMamba.py:11` needs `timm.models.layers.{DropPath,to_2tuple,trunc_normal_}`
    -> a tiny stub; `MedMamba.selective_scan_fn := selective_scan_ref`.
  * bytecode writing is disabled so nothing is dropped into /root/reference.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import sys
sys.dont_write_bytecode = True
import os, json, types, importlib.util
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def _load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    sys.modules.setdefault("selective_scan_cuda", types.ModuleType("selective_scan_cuda"))
    ssi = _load_file("ref_selective_scan_interface",
                     f"{REF}/CrossMamba/FusionMamba/mamba_ssm/ops/selective_scan_interface.py")

    # timm stub (timm is not installed in this image)
    timm = types.ModuleType("timm"); models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__(); self.drop_prob = drop_prob; self.scale_by_keep = scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                mask.div_(keep)
            return x * mask

    layers.DropPath = DropPath
    layers.to_2tuple = lambda v: (v, v) if not isinstance(v, tuple) else v
    layers.trunc_normal_ = nn.init.trunc_normal_
    sys.modules["timm"] = timm; sys.modules["timm.models"] = models
    sys.modules["timm.models.layers"] = layers
    mm = _load_file("ref_MedMamba", f"{REF}/MedMamba.py")
    mm.selective_scan_fn = ssi.selective_scan_ref
    return ssi, mm


def np32(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------------------------
# 1. operator-level vectors: selective_scan_ref forward + autograd grads
#    input distributions follow the reference's own test (test_selective_scan.py:406-441,474)
# ----------------------------------------------------------------------------------------------
def scan_case(ssi, name, batch, dim, dstate, seqlen, groups, has_D=True, has_bias=True, softplus=True,
              has_z=False, seed=0):
    torch.manual_seed(seed)
    A = (-0.5 * torch.rand(dim, dstate)).requires_grad_()
    if groups == 0:      # 3-D B/C  (B, N, L)
        Bm = torch.randn(batch, dstate, seqlen).requires_grad_()
        Cm = torch.randn(batch, dstate, seqlen).requires_grad_()
    else:
        Bm = torch.randn(batch, groups, dstate, seqlen).requires_grad_()
        Cm = torch.randn(batch, groups, dstate, seqlen).requires_grad_()
    D = torch.randn(dim).requires_grad_() if has_D else None
    z = torch.randn(batch, dim, seqlen).requires_grad_() if has_z else None
    bias = (0.5 * torch.rand(dim)).requires_grad_() if has_bias else None
    u = torch.randn(batch, dim, seqlen).requires_grad_()
    delta = (0.5 * torch.rand(batch, dim, seqlen)).requires_grad_()
    out, last = ssi.selective_scan_ref(u, delta, A, Bm, Cm, D, z=z, delta_bias=bias,
                                       delta_softplus=softplus, return_last_state=True)
    g = torch.randn_like(out)
    out.backward(g)
    d = dict(u=np32(u), delta=np32(delta), A=np32(A), B=np32(Bm), C=np32(Cm), g=np32(g),
             out=np32(out), last_state=np32(last),
             du=np32(u.grad), ddelta=np32(delta.grad), dA=np32(A.grad), dB=np32(Bm.grad), dC=np32(Cm.grad),
             softplus=np.array(int(softplus)))
    if has_D:
        d.update(D=np32(D), dD=np32(D.grad))
    if has_bias:
        d.update(delta_bias=np32(bias), ddelta_bias=np32(bias.grad))
    if has_z:
        d.update(z=np32(z), dz=np32(z.grad))
    np.savez_compressed(os.path.join(OUT, f"scan_{name}.npz"), **d)
    print("scan", name, {k: v.shape for k, v in d.items() if k in ("u", "B")})


# ----------------------------------------------------------------------------------------------
# 2. cross-scan / cross-merge permutations captured from SS2D.forward_corev0 (MedMamba.py:386-424)
# ----------------------------------------------------------------------------------------------
def cross_case(mm, ssi, H, W):
    torch.manual_seed(0)
    Bsz, d_model = 2, 2
    blk = mm.SS2D(d_model=d_model, d_state=2)
    D = blk.d_inner
    cap = {}

    def spy(u, delta, A, Bm, Cm, Dp, z=None, delta_bias=None, delta_softplus=False, return_last_state=False):
        cap["xs"] = u.detach().clone()
        # return a tensor whose values identify (b, k*D+d, l) exactly
        return torch.arange(u.numel(), dtype=torch.float32).view_as(u)

    mm.selective_scan_fn = spy
    x = torch.arange(Bsz * D * H * W, dtype=torch.float32).view(Bsz, D, H, W)
    y1, y2, y3, y4 = blk.forward_corev0(x)
    mm.selective_scan_fn = ssi.selective_scan_ref
    xs = cap["xs"].view(Bsz, 4, D, H * W)
    np.savez_compressed(os.path.join(OUT, f"cross_{H}x{W}.npz"),
                        x=np32(x).astype(np.int64), xs=np32(xs).astype(np.int64),
                        y1=np32(y1).astype(np.int64), y2=np32(y2).astype(np.int64),
                        y3=np32(y3).astype(np.int64), y4=np32(y4).astype(np.int64))
    print("cross", H, W)


# ----------------------------------------------------------------------------------------------
# 3. module-level vectors
# ----------------------------------------------------------------------------------------------
def sd_np(mod):
    return {"sd." + k: np32(v) for k, v in mod.state_dict().items()}


def grads_np(mod):
    return {"grad." + k: np32(p.grad) for k, p in mod.named_parameters() if p.grad is not None}


def ss2d_case(mm, name, d_model, d_state, H, W, batch=2, seed=0):
    torch.manual_seed(seed)
    blk = mm.SS2D(d_model=d_model, d_state=d_state)
    # make the conv/proj weights non-trivial but deterministic
    x = torch.randn(batch, H, W, d_model, requires_grad=True)
    y = blk(x)
    g = torch.randn_like(y)
    y.backward(g)
    d = dict(x=np32(x), y=np32(y), g=np32(g), dx=np32(x.grad),
             meta=np.array([d_model, d_state, H, W, batch]))
    d.update(sd_np(blk)); d.update(grads_np(blk))
    np.savez_compressed(os.path.join(OUT, f"ss2d_{name}.npz"), **d)
    print("ss2d", name, list(blk.state_dict().keys()))


def block_case(mm, name, hidden, H, W, batch=2, seed=0):
    torch.manual_seed(seed)
    blk = mm.SS_Conv_SSM(hidden_dim=hidden, drop_path=0.0)
    blk.train()
    x = torch.randn(batch, H, W, hidden, requires_grad=True)
    y = blk(x)
    g = torch.randn_like(y)
    y.backward(g)
    d = dict(x=np32(x), y=np32(y), g=np32(g), dx=np32(x.grad), meta=np.array([hidden, H, W, batch]))
    d.update(sd_np(blk)); d.update(grads_np(blk))   # state dict AFTER the train-mode forward (BN stats updated)
    np.savez_compressed(os.path.join(OUT, f"block_{name}.npz"), **d)
    print("block", name)


def vssm_case(mm, name, depths, dims, num_classes, res, batch=2, seed=0):
    torch.manual_seed(seed)
    net = mm.VSSM(depths=depths, dims=dims, num_classes=num_classes, drop_path_rate=0.0)
    net.train()
    sd0 = {"sd." + k: np32(v).copy() for k, v in net.state_dict().items()}
    x = torch.randn(batch, 3, res, res)
    labels = torch.randint(0, num_classes, (batch,))
    logits = net(x)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    d = dict(x=np32(x), labels=labels.numpy(), logits=np32(logits), loss=np32(loss),
             meta=np.array(list(depths) + list(dims) + [num_classes, res, batch]))
    d.update(sd0)
    # keep the fixture small: only a few representative grads
    keep = ("patch_embed.proj.weight", "head.weight", "layers.0.blocks.0.self_attention.A_logs",
            "layers.0.blocks.0.self_attention.x_proj_weight", "layers.0.blocks.0.self_attention.dt_projs_bias",
            "layers.0.blocks.0.self_attention.Ds", "layers.0.blocks.0.self_attention.in_proj.weight",
            "layers.0.downsample.reduction.weight")
    for k, p in net.named_parameters():
        if k in keep:
            d["grad." + k] = np32(p.grad)
    np.savez_compressed(os.path.join(OUT, f"vssm_{name}.npz"), **d)
    print("vssm", name, float(loss))


def keys_case(mm):
    out = {}
    for tag, kw in (("T", dict()), ("B", dict(depths=[2, 2, 12, 2], dims=[128, 256, 512, 1024]))):
        net = mm.VSSM(num_classes=8, **kw)
        out[tag] = {"n_params": sum(p.numel() for p in net.parameters()),
                    "keys": {k: list(v.shape) for k, v in net.state_dict().items()},
                    "no_weight_decay": sorted(n for n, p in net.named_parameters()
                                              if getattr(p, "_no_weight_decay", False))}
        print("keys", tag, out[tag]["n_params"], len(out[tag]["keys"]))
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


def init_stats_case(mm):
    """Summary statistics of SS2D's special inits (MedMamba.py:329-384) for the init test."""
    torch.manual_seed(0)
    blk = mm.SS2D(d_model=96)
    st = dict(A_logs=np32(blk.A_logs), Ds=np32(blk.Ds),
              dt_bias_softplus_min=float(F.softplus(blk.dt_projs_bias).min()),
              dt_bias_softplus_max=float(F.softplus(blk.dt_projs_bias).max()),
              dt_w_absmax=float(blk.dt_projs_weight.abs().max()), dt_rank=blk.dt_rank, d_inner=blk.d_inner)
    np.savez_compressed(os.path.join(OUT, "ss2d_init.npz"), **{k: np.asarray(v) for k, v in st.items()})


def main():
    torch.set_num_threads(8)
    ssi, mm = import_reference()
    # operator cases: (batch, dim, dstate, L, groups)
    scan_case(ssi, "L49_g4", 2, 8, 16, 49, 4)
    scan_case(ssi, "L64_g4", 2, 8, 16, 64, 4, seed=1)
    scan_case(ssi, "L196_g4", 1, 12, 16, 196, 4, seed=2)
    scan_case(ssi, "L300_g2", 2, 6, 16, 300, 2, seed=3)
    scan_case(ssi, "L130_g1_n4", 2, 5, 4, 130, 1, seed=4)
    scan_case(ssi, "L77_3d", 2, 4, 16, 77, 0, seed=5)
    scan_case(ssi, "L50_noD_nobias", 2, 8, 16, 50, 4, has_D=False, has_bias=False, seed=6)
    scan_case(ssi, "L50_nosoftplus", 2, 8, 16, 50, 4, softplus=False, seed=7)
    scan_case(ssi, "L60_z", 2, 8, 8, 60, 2, has_z=True, seed=8)
    scan_case(ssi, "L1_g4", 2, 8, 16, 1, 4, seed=9)
    scan_case(ssi, "L2100_g4", 1, 4, 16, 2100, 4, seed=10)      # crosses the reference's 2048 chunk
    for hw in ((3, 5), (4, 4), (7, 2)):
        cross_case(mm, ssi, *hw)
    ss2d_case(mm, "d12_5x7", 12, 16, 5, 7)
    ss2d_case(mm, "d48_8x8", 48, 16, 8, 8, batch=1, seed=1)
    block_case(mm, "h24_6x5", 24, 6, 5)
    vssm_case(mm, "tiny", [1, 1], [16, 32], 3, 32)
    keys_case(mm)
    init_stats_case(mm)


if __name__ == "__main__":
    main()
