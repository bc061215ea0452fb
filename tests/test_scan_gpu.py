"""GPU parity tests of the selective-scan operator (HIP kernels through the C ABI) against
  (1) the golden vectors produced by running the reference (tests/golden/scan_*.npz),
  (2) the CPU oracle (oracle/scan_oracle.c) on seeded inputs with the reference's input distributions
      (test_selective_scan.py:406-441) at the model's real per-stage shapes,
  (3) size-independent properties at BASELINE.json's full size (bs 64, stage 0).

Tolerances: fp32 rows of the reference's own test (test_selective_scan.py:398-401,490-502):
out rtol 6e-4 atol 2e-3; du 2x; ddelta 5x; dA 1e-3/5e-3 ... tightened to the north-star's 1e-3 relative
(max-norm) for the forward.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import scan_oracle as so

pytestmark = pytest.mark.gpu

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "scan_*.npz")))
RTOL, ATOL = 6e-4, 2e-3


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def close(got, want, rtol, atol, what):
    got = got.detach().float().cpu().numpy().astype(np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    bad = np.abs(got - want) - (atol + rtol * np.abs(want))
    assert bad.max(initial=-1) <= 0, (f"{what}: max|diff|={np.abs(got - want).max():.3e} "
                                      f"max|ref|={np.abs(want).max():.3e} n_bad={(bad > 0).sum()}")


def relmax(got, want):
    got = got.detach().float().cpu().numpy().astype(np.float64)
    return np.abs(got - np.asarray(want, np.float64)).max() / max(np.abs(want).max(), 1e-30)


def run_hip(t, softplus, z=None, return_last_state=True):
    from medical_image_classification_amd import selective_scan_fn
    return selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t.get("D"), z, t.get("delta_bias"),
                             softplus, return_last_state)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[5:-4] for p in GOLD])
def test_golden(path):
    """Gradient tolerances are `rtol x |ref| + atol x max(1, max|ref|)`: the reference test's own fp32 rows (test_selective_scan.py:
    398-401, 490-502) with the absolute term scaled by the tensor's magnitude (its fixed atol assumes O(1) gradients; dA / dD here
    range over orders of magnitude between cases); forward: the reference rows AND the north-star's hard 1e-3 relative bound."""
    g = np.load(path)
    d = dev()
    names = [n for n in ("u", "delta", "A", "B", "C", "D", "delta_bias", "z") if n in g.files]
    t = {n: torch.from_numpy(g[n]).to(d).requires_grad_() for n in names}
    sp = bool(int(g["softplus"]))
    out, last = run_hip(t, sp, z=t.get("z"))
    close(out, g["out"], RTOL, ATOL, "out")
    assert relmax(out, g["out"]) < 1e-3
    close(last, g["last_state"], RTOL, ATOL, "last_state")
    out.backward(torch.from_numpy(g["g"]).to(d))
    scale = lambda k: max(1.0, float(np.abs(g[k]).max()))
    close(t["u"].grad, g["du"], 2 * RTOL, 2 * ATOL * scale("du"), "du")
    close(t["delta"].grad, g["ddelta"], 5 * RTOL, 5 * ATOL * scale("ddelta"), "ddelta")
    close(t["A"].grad, g["dA"], 1e-3, 5e-3 * scale("dA"), "dA")
    close(t["B"].grad, g["dB"], RTOL, ATOL * scale("dB"), "dB")
    close(t["C"].grad, g["dC"], RTOL, ATOL * scale("dC"), "dC")
    if "D" in t:
        close(t["D"].grad, g["dD"], 1e-3, 1e-3 * scale("dD"), "dD")
    if "delta_bias" in t:
        close(t["delta_bias"].grad, g["ddelta_bias"], 1e-3, 1e-3 * scale("ddelta_bias"), "ddelta_bias")
    if "z" in t:
        close(t["z"].grad, g["dz"], RTOL, ATOL * scale("dz"), "dz")


def make_inputs(batch, dim, N, L, G, seed, R=None):
    """Reference test distributions (test_selective_scan.py:406-441).  With R given, B and C are strided
    VIEWS of one (batch, G, R+2N, L) tensor exactly as SS2D passes them (MedMamba.py:399,405-406)."""
    gen = torch.Generator().manual_seed(seed)
    A = -0.5 * torch.rand(dim, N, generator=gen)
    if R is None:
        Bm = torch.randn(batch, G, N, L, generator=gen)
        Cm = torch.randn(batch, G, N, L, generator=gen)
    else:
        xdbl = torch.randn(batch, G, R + 2 * N, L, generator=gen)
        Bm, Cm = xdbl[:, :, R:R + N], xdbl[:, :, R + N:]
    D = torch.randn(dim, generator=gen)
    bias = 0.5 * torch.rand(dim, generator=gen)
    u = torch.randn(batch, dim, L, generator=gen)
    delta = 0.5 * torch.rand(batch, dim, L, generator=gen)
    g = torch.randn(batch, dim, L, generator=gen)
    return dict(u=u, delta=delta, A=A, B=Bm, C=Cm, D=D, delta_bias=bias), g


# (batch, 4*d_inner, L, R) of the four MedMamba-T stages, batch reduced to 2,
# plus ragged / tiny / odd-group shapes.
STAGE_SHAPES = [(2, 384, 3136, 3), (2, 768, 784, 6), (2, 1536, 196, 12), (2, 3072, 49, 24),
                (1, 512, 1024, 4), (3, 40, 33, None), (1, 4, 1, None), (2, 260, 95, None)]


@pytest.mark.parametrize("shape", STAGE_SHAPES, ids=[f"b{s[0]}_d{s[1]}_L{s[2]}" for s in STAGE_SHAPES])
def test_vs_oracle_model_shapes(shape):
    """Gradient tolerances are `rtol x |ref| + atol x max(1, max|ref|)` (the reference test's fp32 rows, absolute term scaled to the
    tensor's magnitude), PLUS a hard relative-to-max bound of 2e-3 on du / ddelta / dB / dC at the end; forward 1e-3 relative."""
    batch, dim, L, R = shape
    t_cpu, g = make_inputs(batch, dim, 16, L, 4, seed=dim + L, R=R)
    d = dev()
    t = {k: v.to(d) for k, v in t_cpu.items()}          # .to() keeps the strides of the B/C views
    if R is not None:
        xdbl = torch.empty(batch, 4, R + 32, L, device=d)
        xdbl[:, :, R:R + 16] = t["B"]; xdbl[:, :, R + 16:] = t["C"]
        t["B"], t["C"] = xdbl[:, :, R:R + 16], xdbl[:, :, R + 16:]
        assert not t["B"].is_contiguous()
    t = {k: v.requires_grad_() for k, v in t.items()}
    out, last = run_hip(t, True)
    npy = {k: v.numpy() for k, v in t_cpu.items()}
    ref_out, ref_last = so.scan_fwd(npy["u"], npy["delta"], npy["A"], npy["B"], npy["C"], npy["D"], None,
                                    npy["delta_bias"], True)
    close(out, ref_out, RTOL, ATOL, "out")
    assert relmax(out, ref_out) < 1e-3, "north-star forward tolerance (1e-3 rel)"
    close(last, ref_last, RTOL, ATOL, "last_state")
    out.backward(g.to(d))
    ref = so.scan_bwd(npy["u"], npy["delta"], npy["A"], npy["B"], npy["C"], npy["D"], None, npy["delta_bias"],
                      g.numpy(), True)
    sc = lambda a: max(1.0, float(np.abs(a).max()))
    close(t["u"].grad, ref["du"], 2 * RTOL, 2 * ATOL * sc(ref["du"]), "du")
    close(t["delta"].grad, ref["ddelta"], 5 * RTOL, 5 * ATOL * sc(ref["ddelta"]), "ddelta")
    close(t["A"].grad, ref["dA"], 1e-3, 5e-3 * sc(ref["dA"]), "dA")
    close(t["B"].grad, ref["dB"], RTOL, ATOL * sc(ref["dB"]), "dB")
    close(t["C"].grad, ref["dC"], RTOL, ATOL * sc(ref["dC"]), "dC")
    close(t["D"].grad, ref["dD"], 1e-3, 1e-3 * sc(ref["dD"]), "dD")
    close(t["delta_bias"].grad, ref["ddelta_bias"], 1e-3, 1e-3 * sc(ref["ddelta_bias"]), "ddelta_bias")
    for k in ("u", "delta", "B", "C"):
        assert relmax(t[k].grad, ref["d" + k]) < 2e-3, k


@pytest.mark.parametrize("N,G", [(1, 1), (2, 2), (3, 1), (4, 4), (8, 2), (12, 1), (16, 1)])
def test_state_sizes(N, G):
    t_cpu, g = make_inputs(2, 8 * G, N, 70, G, seed=N)
    d = dev()
    t = {k: v.to(d).requires_grad_() for k, v in t_cpu.items()}
    out, last = run_hip(t, True)
    npy = {k: v.numpy() for k, v in t_cpu.items()}
    ref_out, ref_last = so.scan_fwd(npy["u"], npy["delta"], npy["A"], npy["B"], npy["C"], npy["D"], None,
                                    npy["delta_bias"], True)
    close(out, ref_out, RTOL, ATOL, "out"); close(last, ref_last, RTOL, ATOL, "last")
    out.backward(g.to(d))
    ref = so.scan_bwd(npy["u"], npy["delta"], npy["A"], npy["B"], npy["C"], npy["D"], None, npy["delta_bias"],
                      g.numpy(), True)
    for k in ("u", "delta", "A", "B", "C", "D", "delta_bias"):
        close(t[k].grad, ref["d" + k], 3e-3, 1e-2 * max(1.0, float(np.abs(ref["d" + k]).max())), "d" + k)


def test_channel_last_layout():
    """u/delta/dout given as (B,L,D)-contiguous tensors viewed as (B,D,L): same results as the (B,D,L) layout."""
    t_cpu, g = make_inputs(2, 192, 16, 100, 4, seed=5)
    d = dev()
    a = {k: v.to(d).requires_grad_() for k, v in t_cpu.items()}
    b = {k: v.to(d) for k, v in t_cpu.items()}
    for k in ("u", "delta"):
        b[k] = b[k].transpose(1, 2).contiguous().transpose(1, 2)
        assert b[k].stride(1) == 1
    b = {k: v.requires_grad_() for k, v in b.items()}
    oa, _ = run_hip(a, True); ob, _ = run_hip(b, True)
    gd = g.to(d)
    oa.backward(gd); ob.backward(gd.transpose(1, 2).contiguous().transpose(1, 2))
    assert torch.allclose(oa, ob, rtol=1e-6, atol=1e-6)
    for k in ("u", "delta"):
        assert torch.allclose(a[k].grad, b[k].grad, rtol=1e-5, atol=1e-5), k
    for k in ("A", "B", "C", "D", "delta_bias"):
        assert torch.allclose(a[k].grad, b[k].grad, rtol=1e-4, atol=1e-4), k


def test_half_dtypes_and_constant_bc():
    from medical_image_classification_amd import selective_scan_fn
    d = dev()
    t_cpu, _ = make_inputs(2, 8, 4, 40, 1, seed=9)
    ref_out, _ = so.scan_fwd(*[t_cpu[k].numpy() for k in ("u", "delta", "A", "B", "C", "D")], None,
                             t_cpu["delta_bias"].numpy(), True)
    for dt, tol in ((torch.float16, 5e-3), (torch.bfloat16, 5e-2)):
        out = selective_scan_fn(t_cpu["u"].to(d, dt), t_cpu["delta"].to(d, dt), t_cpu["A"].to(d),
                                t_cpu["B"].to(d, dt), t_cpu["C"].to(d, dt), t_cpu["D"].to(d), None,
                                t_cpu["delta_bias"].to(d), True)
        assert out.dtype == dt
        u16 = t_cpu["u"].to(dt).float().numpy(); d16 = t_cpu["delta"].to(dt).float().numpy()
        r16, _ = so.scan_fwd(u16, d16, t_cpu["A"].numpy(), t_cpu["B"].to(dt).float().numpy(),
                             t_cpu["C"].to(dt).float().numpy(), t_cpu["D"].numpy(), None, t_cpu["delta_bias"].numpy(), True)
        close(out, r16, tol * 6, tol, f"out {dt}")
    # constant (dim, N) B and C  (selective_scan_ref: is_variable_B False, selective_scan_interface.py:128-129,141-142)
    Bc, Cc = torch.randn(8, 4), torch.randn(8, 4)
    Bexp = Bc.view(1, 8, 4, 1).expand(2, 8, 4, 40).contiguous().numpy()
    Cexp = Cc.view(1, 8, 4, 1).expand(2, 8, 4, 40).contiguous().numpy()
    ref, _ = so.scan_fwd(t_cpu["u"].numpy(), t_cpu["delta"].numpy(), t_cpu["A"].numpy(), Bexp, Cexp, None, None, None, False)
    Bd, Cd = Bc.to(d).requires_grad_(), Cc.to(d).requires_grad_()
    out = selective_scan_fn(t_cpu["u"].to(d), t_cpu["delta"].to(d), t_cpu["A"].to(d), Bd, Cd)
    close(out, ref, RTOL, ATOL, "const B/C")
    out.sum().backward()
    assert Bd.grad.shape == (8, 4) and Cd.grad.shape == (8, 4)


def test_errors():
    from medical_image_classification_amd import selective_scan_fn
    d = dev()
    u = torch.randn(2, 8, 16, device=d); A = -torch.rand(8, 4, device=d); B = torch.randn(2, 4, 16, device=d)
    with pytest.raises(RuntimeError):
        selective_scan_fn(u, u[:, :4], A, B, B)
    with pytest.raises(RuntimeError):
        selective_scan_fn(u, u, A, B[:, :3], B)
    with pytest.raises(RuntimeError):
        selective_scan_fn(u, u, torch.complex(A, A), B, B)
    with pytest.raises(RuntimeError, match="state dimension"):
        selective_scan_fn(u, u, -torch.rand(8, 300, device=d), torch.randn(2, 300, 16, device=d), torch.randn(2, 300, 16, device=d))
    out = selective_scan_fn(u[:0], u[:0], A, B[:0], B[:0])          # empty batch
    assert out.shape == (0, 8, 16)


def test_full_size_properties():
    """BASELINE config 2 stage-0 size (bs 64: u (64,384,3136)): properties that need no oracle run.
    (a) linearity in u for fixed delta: scan(a*u1 + u2) = a*scan(u1) + scan(u2);
    (b) rows are independent of the batch they sit in: a slice equals the small run checked against the oracle;
    (c) reversal-free causality: changing u at l >= l* leaves out[..., :l*] bit-identical."""
    d = dev()
    batch, dim, L, N, G = 64, 384, 3136, 16, 4
    gen = torch.Generator(device=d).manual_seed(0)
    A = -0.5 * torch.rand(dim, N, device=d, generator=gen)
    Bm = torch.randn(batch, G, N, L, device=d, generator=gen)
    Cm = torch.randn(batch, G, N, L, device=d, generator=gen)
    D = torch.randn(dim, device=d, generator=gen)
    bias = 0.5 * torch.rand(dim, device=d, generator=gen)
    u1 = torch.randn(batch, dim, L, device=d, generator=gen)
    u2 = torch.randn(batch, dim, L, device=d, generator=gen)
    delta = 0.5 * torch.rand(batch, dim, L, device=d, generator=gen)
    from medical_image_classification_amd import selective_scan_fn
    f = lambda u: selective_scan_fn(u, delta, A, Bm, Cm, D, None, bias, True)
    o1, o2 = f(u1), f(u2)
    o12 = f(2.5 * u1 + u2)
    lin = 2.5 * o1 + o2
    assert (o12 - lin).abs().max().item() <= 1e-3 * lin.abs().max().item()
    # (b) one batch element against the oracle
    sl = slice(7, 8)
    ref, _ = so.scan_fwd(u1[sl].cpu().numpy(), delta[sl].cpu().numpy(), A.cpu().numpy(), Bm[sl].cpu().numpy(),
                         Cm[sl].cpu().numpy(), D.cpu().numpy(), None, bias.cpu().numpy(), True)
    close(o1[sl], ref, RTOL, ATOL, "full-size slice")
    # (c) causality, bit exact
    u3 = u1.clone(); u3[:, :, 2000:] = 0
    o3 = f(u3)
    assert torch.equal(o3[:, :, :2000], o1[:, :, :2000])


def test_long_sequence_medmamba_b_stage0():
    """BASELINE config 3 stage 0 (MedMamba-B @512x512): L = 128*128 = 16384 (512 chunks), D = 128 -> dim 512, one batch
    element; forward and all gradients vs the oracle.  With the test distributions the states reach |h| ~ 4e2 over 16k
    steps and both sides are fp32 with different rounding (sequential expf loop vs exp2 + different association), so the
    bound here is the north-star's max-norm relative 1e-3 (measured: 6e-5 forward), not the short-sequence absolute floor."""
    batch, dim, L, R = 1, 512, 16384, 4
    t_cpu, g = make_inputs(batch, dim, 16, L, 4, seed=77, R=R)
    d = dev()
    t = {k: v.to(d).requires_grad_() for k, v in t_cpu.items()}
    out, last = run_hip(t, True)
    npy = {k: v.numpy() for k, v in t_cpu.items()}
    ref_out, ref_last = so.scan_fwd(npy["u"], npy["delta"], npy["A"], npy["B"], npy["C"], npy["D"], None, npy["delta_bias"], True)
    assert relmax(out, ref_out) < 1e-3 and relmax(last, ref_last) < 1e-3
    out.backward(g.to(d))
    ref = so.scan_bwd(npy["u"], npy["delta"], npy["A"], npy["B"], npy["C"], npy["D"], None, npy["delta_bias"], g.numpy(), True)
    for k in ("u", "delta", "A", "B", "C", "D", "delta_bias"):
        assert relmax(t[k].grad, ref["d" + k]) < 2e-3, (k, relmax(t[k].grad, ref["d" + k]))


@pytest.mark.parametrize("cfg", [(2, 128, 16, 100, 1), (1, 40, 12, 70, 2), (2, 64, 5, 33, 1), (1, 256, 16, 49, 4)])
def test_scalar_decay_kernels_match_dense_A(cfg):
    """A given as a stride-0 broadcast over the state axis (one decay rate per channel, the SSD / Mamba-2 form) takes the
    scalar-decay kernel variants (one exp2 per position); results and every gradient must equal the dense-A kernels and
    the oracle.  Channel-last activations, as cnn_mamba.mamba_chunk_scan_combined passes them."""
    batch, dim, N, L, G = cfg
    t_cpu, g = make_inputs(batch, dim, N, L, G, seed=21)
    a_col = t_cpu["A"][:, :1].clone()                              # (dim, 1): one rate per channel
    t_cpu["A"] = a_col.expand(dim, N).contiguous()
    ref_out, _ = so.scan_fwd(*[t_cpu[k].numpy() for k in ("u", "delta", "A", "B", "C", "D")], None,
                             t_cpu["delta_bias"].numpy(), True)
    d = dev()
    cl = lambda v: v.transpose(1, 2).contiguous().transpose(1, 2)   # (B,D,L) view of a channel-last tensor
    res = {}
    for mode in ("dense", "scalar"):
        t = {k: v.to(d) for k, v in t_cpu.items()}
        t["u"], t["delta"] = cl(t["u"]), cl(t["delta"])
        col = a_col.to(d).requires_grad_()
        t = {k: v.requires_grad_() for k, v in t.items() if k != "A"}
        t["A"] = col.expand(dim, N) if mode == "scalar" else col.expand(dim, N).contiguous()
        assert (t["A"].stride(1) == 0) == (mode == "scalar")
        out, _ = run_hip(t, True)
        out.backward(cl(g.to(d)))
        res[mode] = (out.detach(), {k: v.grad for k, v in t.items() if k != "A"}, col.grad)
    close(res["scalar"][0], ref_out, RTOL, ATOL, "out vs oracle")
    assert torch.allclose(res["scalar"][0], res["dense"][0], rtol=1e-5, atol=1e-5)
    for k in res["dense"][1]:
        a, b = res["scalar"][1][k], res["dense"][1][k]
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * max(1.0, float(b.abs().max()))), k
    a, b = res["scalar"][2], res["dense"][2]
    assert torch.allclose(a, b, rtol=1e-3, atol=1e-3 * max(1.0, float(b.abs().max()))), "dA"


# The reference's own parametrisation (mamba_ssm/ops/test_selective_scan.py:372-392): dim 768, batch 2, dstate 1, variable
# B and C in 1 or 2 groups, seqlen 64 ... 4096, every combination of D / delta_bias / delta_softplus, three I/O dtypes --
# with its tolerances (:398-401 forward, :490-502 gradients), against the pinned oracle instead of selective_scan_ref.
REF_FLAGS = [(g, hd, hb, sp) for g in (1, 2) for hd in (False, True) for hb in (False, True) for sp in (False, True)]


@pytest.mark.parametrize("seqlen", [64, 128, 256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("itype", [torch.float32, torch.float16, torch.bfloat16], ids=["fp32", "fp16", "bf16"])
def test_reference_test_grid(seqlen, itype):
    from medical_image_classification_amd import selective_scan_fn
    batch, dim, dstate = 2, 768, 1
    rtol, atol = (6e-4, 2e-3) if itype == torch.float32 else (3e-3, 5e-3)
    if itype == torch.bfloat16:
        rtol, atol = 3e-2, 5e-2
    rtolw, atolw = 1e-3, 1e-3
    d = dev()
    flags = REF_FLAGS if itype == torch.float32 else REF_FLAGS[-3:]          # half precisions: a few combinations each
    for groups, has_D, has_bias, softplus in flags:
        gen = torch.Generator().manual_seed(0)
        A = -0.5 * torch.rand(dim, dstate, generator=gen)
        Bm = torch.randn(batch, groups, dstate, seqlen, generator=gen).to(itype)
        Cm = torch.randn(batch, groups, dstate, seqlen, generator=gen).to(itype)
        D = torch.randn(dim, generator=gen) if has_D else None
        bias = 0.5 * torch.rand(dim, generator=gen) if has_bias else None
        u = torch.randn(batch, dim, seqlen, generator=gen).to(itype)
        delta = (0.5 * torch.rand(batch, dim, seqlen, generator=gen)).to(itype)
        g = torch.randn(batch, dim, seqlen, generator=gen).to(itype)
        f32 = lambda t: None if t is None else t.float().numpy()
        out_ref, last_ref = so.scan_fwd(f32(u), f32(delta), f32(A), f32(Bm), f32(Cm), f32(D), None, f32(bias), softplus)
        gr = so.scan_bwd(f32(u), f32(delta), f32(A), f32(Bm), f32(Cm), f32(D), None, f32(bias), f32(g), softplus)
        t = {k: (v.to(d).requires_grad_() if v is not None else None)
             for k, v in dict(u=u, delta=delta, A=A, B=Bm, C=Cm, D=D, bias=bias).items()}
        out, last = selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], None, t["bias"], softplus, True)
        assert out.dtype == itype
        tag = f"L={seqlen} g={groups} D={has_D} bias={has_bias} sp={softplus}"
        close(out, out_ref, rtol, atol, "out " + tag)
        close(last, last_ref, rtol, atol, "state " + tag)
        out.backward(g.to(d))
        close(t["u"].grad, gr["du"], 2 * rtol, 2 * atol, "du " + tag)
        close(t["delta"].grad, gr["ddelta"], 5 * rtol, 10 * atol, "ddelta " + tag)
        close(t["A"].grad, gr["dA"], rtolw, 5 * atolw * max(1.0, float(np.abs(gr["dA"]).max()) / 50), "dA " + tag)
        close(t["B"].grad, gr["dB"], rtol, atol * max(1.0, float(np.abs(gr["dB"]).max()) / 50), "dB " + tag)
        close(t["C"].grad, gr["dC"], rtol, atol * max(1.0, float(np.abs(gr["dC"]).max()) / 50), "dC " + tag)
        if has_D:
            close(t["D"].grad, gr["dD"], rtolw, atolw * max(1.0, float(np.abs(gr["dD"]).max()) / 50), "dD " + tag)
        if has_bias:
            close(t["bias"].grad, gr["ddelta_bias"], rtolw, atolw * max(1.0, float(np.abs(gr["ddelta_bias"]).max()) / 50), "dbias " + tag)


@pytest.mark.parametrize("N", [17, 32, 64, 100, 256])
def test_wide_state_axis_slices(N):
    """dstate up to the reference's 256 (selective_scan.cpp:262): ceil(N / 16) launches over state slices whose outputs and
    du / ddelta add up; output, last state and every gradient vs the C oracle."""
    batch, dim, L, G = 2, 24, 70, 2
    t_cpu, g = make_inputs(batch, dim, N, L, G, seed=N)
    d = dev()
    t = {k: v.to(d).requires_grad_() for k, v in t_cpu.items()}
    out, last = run_hip(t, True)
    npy = {k: v.numpy() for k, v in t_cpu.items()}
    ref_out, ref_last = so.scan_fwd(npy["u"], npy["delta"], npy["A"], npy["B"], npy["C"], npy["D"], None, npy["delta_bias"], True)
    sc = lambda a: max(1.0, float(np.abs(a).max()))
    close(out, ref_out, RTOL, ATOL * sc(ref_out), "out")
    close(last, ref_last, RTOL, ATOL, "last_state")
    out.backward(g.to(d))
    ref = so.scan_bwd(npy["u"], npy["delta"], npy["A"], npy["B"], npy["C"], npy["D"], None, npy["delta_bias"], g.numpy(), True)
    close(t["u"].grad, ref["du"], 2 * RTOL, 2 * ATOL * sc(ref["du"]), "du")
    close(t["delta"].grad, ref["ddelta"], 5 * RTOL, 5 * ATOL * sc(ref["ddelta"]), "ddelta")
    close(t["A"].grad, ref["dA"], 1e-3, 5e-3 * sc(ref["dA"]), "dA")
    close(t["B"].grad, ref["dB"], RTOL, ATOL * sc(ref["dB"]), "dB")
    close(t["C"].grad, ref["dC"], RTOL, ATOL * sc(ref["dC"]), "dC")
    close(t["D"].grad, ref["dD"], 1e-3, 1e-3 * sc(ref["dD"]), "dD")
    close(t["delta_bias"].grad, ref["ddelta_bias"], 1e-3, 1e-3 * sc(ref["ddelta_bias"]), "ddelta_bias")


@pytest.mark.parametrize("const_b", [True, False])
def test_mixed_variable_and_constant_bc(const_b):
    """One of B / C input-dependent (batch, N, L), the other a per-channel constant (dim, N) -- the reference's
    is_variable_B != is_variable_C case (selective_scan.cpp:244-259): gradients come back in the operands' own shapes."""
    from medical_image_classification_amd import selective_scan_fn
    batch, dim, N, L = 2, 12, 8, 40
    gen = torch.Generator().manual_seed(5 + const_b)
    u = torch.randn(batch, dim, L, generator=gen); delta = 0.5 * torch.rand(batch, dim, L, generator=gen)
    A = -0.5 * torch.rand(dim, N, generator=gen)
    var = torch.randn(batch, N, L, generator=gen); con = torch.randn(dim, N, generator=gen)
    g = torch.randn(batch, dim, L, generator=gen)
    d = dev()
    t = [v.to(d).requires_grad_() for v in (u, delta, A, con if const_b else var, var if const_b else con)]
    out = selective_scan_fn(t[0], t[1], t[2], t[3], t[4], None, None, None, True)
    out.backward(g.to(d))
    # oracle on the expanded operands
    Bx = (con.view(1, dim, N, 1).expand(batch, dim, N, L) if const_b else var.view(batch, 1, N, L).expand(batch, dim, N, L)).contiguous()
    Cx = (var.view(batch, 1, N, L).expand(batch, dim, N, L) if const_b else con.view(1, dim, N, 1).expand(batch, dim, N, L)).contiguous()
    ref_out, _ = so.scan_fwd(u.numpy(), delta.numpy(), A.numpy(), Bx.numpy(), Cx.numpy(), None, None, None, True)
    ref = so.scan_bwd(u.numpy(), delta.numpy(), A.numpy(), Bx.numpy(), Cx.numpy(), None, None, None, g.numpy(), True)
    close(out, ref_out, RTOL, ATOL, "out")
    dcon = (ref["dB"] if const_b else ref["dC"]).sum(axis=(0, 3))
    dvar = (ref["dC"] if const_b else ref["dB"]).sum(axis=1)
    assert t[3].grad.shape == t[3].shape and t[4].grad.shape == t[4].shape
    sc = lambda a: max(1.0, float(np.abs(a).max()))
    close(t[3 if const_b else 4].grad, dcon, 1e-3, 2e-3 * sc(dcon), "d const")
    close(t[4 if const_b else 3].grad, dvar, 1e-3, 2e-3 * sc(dvar), "d var")
