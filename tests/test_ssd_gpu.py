"""GPU parity of the SSD (Mamba-2) widening, through the C-ABI scan kernels, against oracle/ssd_oracle.py (sequential
float64 restatement, itself checked against the pinned S6 oracle in tests/test_ssd_cpu.py).  PARITY UNPINNED against the
reference's Triton dependency (mamba_ssm 2.2.2, absent from the tree).  Tolerance: 1e-3 relative to the tensor's max
(north_star's fp32 bound)."""
import numpy as np
import pytest
import torch

from oracle import ssd_oracle

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def close(got, want, tol, msg):
    want = want.detach().float().cpu().numpy()
    np.testing.assert_allclose(got.detach().float().cpu().numpy(), want, rtol=tol,
                               atol=max(1e-6, tol * float(np.abs(want).max())), err_msg=msg)


@pytest.mark.parametrize("cfg", [(2, 30, 8, 64, 1, 64), (1, 49, 4, 16, 1, 16), (2, 65, 6, 8, 2, 24), (1, 7, 2, 64, 1, 5),
                                 (1, 300, 4, 8, 1, 40)])
@pytest.mark.parametrize("hdim_D", [False, True])
@pytest.mark.parametrize("chunked", [False, True])
def test_chunk_scan_combined_vs_restatement(cfg, hdim_D, chunked, monkeypatch):
    """Both evaluations of the operator -- the scan kernels (16 states per launch) and the chunked-GEMM (SSD) form -- against
    the sequential float64 restatement, forward and every gradient."""
    from medical_image_classification_amd import cnn_mamba as cm
    from medical_image_classification_amd.cnn_mamba import mamba_chunk_scan_combined
    monkeypatch.setattr(cm, "SSD_CHUNKED_MIN_STATE", 1 if chunked else 0)
    b, l, h, p, g, n = cfg
    gen = torch.Generator().manual_seed(11)
    mk = lambda *s: torch.randn(*s, generator=gen)
    x, dt, B, C = mk(b, l, h, p), mk(b, l, h), mk(b, l, g, n), mk(b, l, g, n)
    A = -torch.rand(h, generator=gen) * 4 - 0.2
    D = mk(h, p) if hdim_D else mk(h)
    bias = mk(h)
    gy = mk(b, l, h, p)
    cpu = [t.clone().requires_grad_() for t in (x, dt, A, B, C, D, bias)]
    gpu = [t.to(dev()).requires_grad_() for t in (x, dt, A, B, C, D, bias)]
    yr = ssd_oracle.ssd_scan_ref(cpu[0], cpu[1], cpu[2], cpu[3], cpu[4], D=cpu[5], dt_bias=cpu[6], dt_softplus=True)
    yd = mamba_chunk_scan_combined(gpu[0], gpu[1], gpu[2], gpu[3], gpu[4], chunk_size=256, D=gpu[5], z=None,
                                   dt_bias=gpu[6], dt_softplus=True)
    assert yd.shape == (b, l, h, p) and yd.dtype == torch.float32
    yr.backward(gy); yd.backward(gy.to(dev()))
    close(yd, yr, 1e-3, "y")
    for name, a, r in zip(("dx", "ddt", "dA", "dB", "dC", "dD", "dbias"), gpu, cpu):
        close(a.grad, r.grad, 2e-3, name)


def test_chunk_scan_combined_no_softplus_no_bias_no_D():
    from medical_image_classification_amd.cnn_mamba import mamba_chunk_scan_combined
    gen = torch.Generator().manual_seed(5)
    x, dt = torch.randn(1, 20, 2, 8, generator=gen), torch.rand(1, 20, 2, generator=gen) * 0.5
    A, B, C = -torch.ones(2), torch.randn(1, 20, 1, 8, generator=gen), torch.randn(1, 20, 1, 8, generator=gen)
    yr = ssd_oracle.ssd_scan_ref(x, dt, A, B, C)
    yd = mamba_chunk_scan_combined(*(t.to(dev()) for t in (x, dt, A, B, C)), chunk_size=64)
    close(yd, yr, 1e-3, "y")
    with pytest.raises(RuntimeError):
        mamba_chunk_scan_combined(*(t.to(dev()) for t in (x, dt, A, B, C)), chunk_size=64, z=x.to(dev()))


@pytest.mark.parametrize("cfg", [(64, 16, 64, 6, 5, False), (32, 4, 16, 3, 7, True), (64, 16, 64, 1, 1, False)])
def test_ss2d_with_ssd_vs_restatement(cfg):
    from medical_image_classification_amd.cnn_mamba import SS2D_with_SSD
    d_model, d_state, headdim, H, W, hdim_D = cfg
    torch.manual_seed(2)
    m = SS2D_with_SSD(d_model=d_model, d_state=d_state, headdim=headdim, D_has_hdim=hdim_D)
    with torch.no_grad():                                        # break the symmetry of the four copies / ones
        m.Ds.add_(torch.randn_like(m.Ds) * 0.3); m.A_logs.add_(torch.randn_like(m.A_logs) * 0.3)
        m.dt_bias.add_(torch.randn_like(m.dt_bias)); m.norm.weight.add_(torch.randn_like(m.norm.weight) * 0.2)
    ref = SS2D_with_SSD(d_model=d_model, d_state=d_state, headdim=headdim, D_has_hdim=hdim_D)
    ref.load_state_dict(m.state_dict())
    m.to(dev())
    u = torch.randn(2, H, W, d_model)
    g = torch.randn(2, H, W, d_model)
    ur = u.clone().requires_grad_(); ud = u.to(dev()).requires_grad_()
    yr = ssd_oracle.ss2d_ssd_forward_oracle(ref, ur)
    yd = m(ud)
    yr.backward(g); yd.backward(g.to(dev()))
    close(yd, yr, 1e-3, "y")
    close(ud.grad, ur.grad, 2e-3, "du")
    pr = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        close(p.grad, pr[k].grad, 2e-3, k)


def test_ssd_vssm_train_step_vs_cpu_restatement():
    from medical_image_classification_amd.cnn_mamba import VSSM
    torch.manual_seed(4)
    net = VSSM(depths=[1, 1], dims=[128, 256], num_classes=3, drop_path_rate=0.0)
    ref = VSSM(depths=[1, 1], dims=[128, 256], num_classes=3, drop_path_rate=0.0)
    ref.load_state_dict(net.state_dict())
    ssd_oracle.install_ssd(ref)
    net.to(dev()).train(); ref.train()
    x = torch.randn(2, 3, 32, 32)
    t = torch.tensor([0, 2])
    lr_ = torch.nn.functional.cross_entropy(ref(x), t)
    ld_ = torch.nn.functional.cross_entropy(net(x.to(dev())), t.to(dev()))
    lr_.backward(); ld_.backward()
    assert abs(float(ld_) - float(lr_)) <= 1e-3 * max(1.0, abs(float(lr_)))
    pr = dict(ref.named_parameters())
    for k, p in net.named_parameters():
        if "self_attention" in k or k.startswith("head"):
            r = pr[k].grad.numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=1e-2, atol=max(1e-5, 5e-3 * float(np.abs(r).max())),
                                       err_msg=k)


def test_ssd_vssm_default_config_autocast_step_runs():
    """CNN_Mamba.VSSM as train.py:58 builds it (dims 128..1024, d_state 16 -> SSD state 64), bf16 autocast, 64x64 input."""
    from medical_image_classification_amd.cnn_mamba import VSSM
    torch.manual_seed(0)
    net = VSSM(num_classes=7).to(dev()).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    x = torch.randn(2, 3, 64, 64, device=dev())
    t = torch.tensor([1, 5], device=dev())
    losses = []
    for _ in range(2):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(net(x), t)
        loss.backward(); opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


@pytest.mark.parametrize("cfg", [(64, 16, 64, 6, 5), (32, 4, 16, 3, 7)])
def test_crossmamba_vs_restatement(cfg):
    from medical_image_classification_amd.crossmamba import CrossMamba
    d_model, d_state, headdim, H, W = cfg
    torch.manual_seed(6)
    m = CrossMamba(d_model=d_model, d_state=d_state, headdim=headdim)
    with torch.no_grad():
        m.Ds.add_(torch.randn_like(m.Ds) * 0.3); m.A_logs.add_(torch.randn_like(m.A_logs) * 0.3)
        m.dt_bias.add_(torch.randn_like(m.dt_bias)); m.norm.weight.add_(torch.randn_like(m.norm.weight) * 0.2)
    ref = CrossMamba(d_model=d_model, d_state=d_state, headdim=headdim)
    ref.load_state_dict(m.state_dict())
    m.to(dev())
    us = [torch.randn(2, H, W, d_model) for _ in range(4)]
    gs = [torch.randn(2, H, W, d_model) for _ in range(2)]
    ur = [u.clone().requires_grad_() for u in us]
    ud = [u.to(dev()).requires_grad_() for u in us]
    r1, r2 = ssd_oracle.crossmamba_forward_oracle(ref, *ur)
    d1, d2 = m(*ud)
    (r1 * gs[0] + r2 * gs[1]).sum().backward()
    (d1 * gs[0].to(dev()) + d2 * gs[1].to(dev())).sum().backward()
    close(d1, r1, 1e-3, "out1"); close(d2, r2, 1e-3, "out2")
    for i in range(4):
        close(ud[i].grad, ur[i].grad, 2e-3, f"du{i}")
    pr = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        if pr[k].grad is None:                               # in_proj / conv2d: constructed, never used (as in the reference)
            assert p.grad is None, k
        else:
            close(p.grad, pr[k].grad, 2e-3, k)


@pytest.mark.parametrize("cfg", [(64, 16, 64, 9, 7, False), (32, 8, 16, 5, 12, True), (32, 40, 16, 5, 6, False),
                                 (64, 128, 64, 4, 5, False)])
def test_ssd_pixel_order_kernels_match_gathered_path(cfg, monkeypatch):
    """ssd_scan_merge through the pixel-order kernels (MS_SCAN_BC_MAP / MS_SCAN_ACCUMULATE, 4 x ceil(N/16) launches on the conv
    output itself) and through the gathered sequences + chunked GEMMs == the gathered-copies path on the plain scan operator:
    output and every gradient."""
    from medical_image_classification_amd import cnn_mamba as cm
    d_model, d_state, headdim, H, W, hdim_D = cfg
    torch.manual_seed(8)
    m = cm.SS2D_with_SSD(d_model=d_model, d_state=d_state, headdim=headdim, D_has_hdim=hdim_D).to(dev())
    with torch.no_grad():
        m.Ds.add_(torch.randn_like(m.Ds) * 0.3); m.A_logs.add_(torch.randn_like(m.A_logs) * 0.3)
        m.dt_bias.add_(torch.randn_like(m.dt_bias))
    u = torch.randn(2, H, W, d_model, device=dev())
    g = torch.randn(2, H, W, d_model, device=dev())
    res = {}
    for name, pixel, min_state in (("pixel", True, 0), ("gathered", False, 0), ("chunked", False, 1)):
        monkeypatch.setattr(cm, "SSD_PIXEL_ORDER", pixel)
        monkeypatch.setattr(cm, "SSD_CHUNKED_MIN_STATE", min_state)
        m.zero_grad(set_to_none=True)
        ui = u.clone().requires_grad_()
        y = m(ui)
        y.backward(g)
        res[name] = (y.detach(), ui.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()})
    y0, du0, p0 = res["gathered"]
    for name in ("pixel", "chunked"):
        y1, du1, p1 = res[name]
        close(y1, y0, 1e-4, f"{name} y")
        close(du1, du0, 1e-3, f"{name} du")
        for k in p0:
            close(p1[k], p0[k], 1e-3, f"{name} {k}")


def test_vfefm_small_matches_cpu_oracle():
    """VFEFM (CrossMamba_fusion_2b2.py:1078-1285; two encoders, CrossMamba fusions, skip decoder, patch expands) on the HIP
    path == the same weights on CPU through the oracle modules: fused image, loss and parameter gradients.  PARITY UNPINNED
    against the reference's Triton dependency; the CPU side is the pinned S6 oracle by expansion."""
    from medical_image_classification_amd.crossmamba import VFEFM
    from medical_image_classification_amd.fusion_loss import FusionLoss
    cfg = dict(depths=[1, 1, 1, 1], dims=[64, 128, 256, 512], depths_decoder=[1, 1, 1, 1], dims_decoder=[512, 256, 128, 64],
               d_state=20, drop_path_rate=0.0)
    torch.manual_seed(11)
    net, ref = VFEFM(**cfg), VFEFM(**cfg)
    ref.load_state_dict(net.state_dict())
    ssd_oracle.install_ssd(ref)
    net.to(dev()).train(); ref.train()
    x1, x2 = torch.rand(1, 3, 64, 64), torch.rand(1, 3, 64, 64)          # (the CPU oracle side dominates this test's time)
    out_r = ssd_oracle.vfefm_forward_oracle(ref, x1, x2)
    out_d = net(x1.to(dev()), x2.to(dev()))
    assert tuple(out_d.shape) == (1, 1, 64, 64)
    close(out_d, out_r, 2e-3, "fused image")
    crit = FusionLoss()
    lr_ = crit(x1, x2, out_r.clamp(0, 1))[0]
    ld_ = crit.to(dev())(x1.to(dev()), x2.to(dev()), out_d.clamp(0, 1))[0]
    lr_.backward(); ld_.backward()
    assert abs(float(ld_) - float(lr_)) <= 1e-3 * max(1.0, abs(float(lr_)))
    pr = dict(ref.named_parameters())
    checked = 0
    for k, p in net.named_parameters():
        if pr[k].grad is None:
            assert p.grad is None, k
            continue
        if ("fusion" in k or "self_attention" in k or "final" in k or "in_proj" in k) and "norm" not in k:
            r = pr[k].grad.numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=2e-2, atol=max(1e-6, 1e-2 * float(np.abs(r).max())), err_msg=k)
            checked += 1
    assert checked > 60


def test_vfefm_default_config_fusion_step_runs():
    """The model CrossMamba/train.py:80-91 builds (dims 128..1024, decoder depths 2/9/2/2, d_state 128 -> SSD state 512),
    one bf16-autocast fusion_step on 1 x 3 x 64 x 64 pairs: finite losses, every used parameter gets a finite gradient."""
    from medical_image_classification_amd.train_fusion import build_fusion_model, fusion_step, synthetic_pair
    from medical_image_classification_amd.fusion_loss import FusionLoss
    from medical_image_classification_amd.train import make_adam
    torch.manual_seed(0)
    net = build_fusion_model().to(dev()).train()
    opt = make_adam(net.parameters(), lr=2e-4)
    vis, ir = synthetic_pair(2, 64, dev())
    before = net.final_conv.weight.detach().clone()
    terms = fusion_step(net, opt, FusionLoss().to(dev()), vis, ir, torch.bfloat16)
    assert all(torch.isfinite(t).item() for t in terms)
    unused = 0
    for n, p in net.named_parameters():
        if p.grad is None:
            unused += 1
            # constructed, never used (as in the reference): CrossMamba's in_proj / conv2d, in_proj1/2 of the skip-less first decoder stage
            assert ".fusion.in_proj." in n or ".fusion.conv2d." in n or n.startswith("layers_up.0.in_proj"), n
        else:
            assert torch.isfinite(p.grad).all().item(), n
    assert unused == 3 * 8 + 4
    assert not torch.equal(before, net.final_conv.weight.detach())


@pytest.mark.parametrize("cfg", [(2, 5, 2, 3, 2, 8), (1, 49, 1, 16, 4, 64), (3, 1, 1, 5, 1, 4), (2, 6, 1, 7, 3, 12)])
def test_chunk_carry_kernel_vs_loop(cfg):
    """ms_ssd_chunk_carry (forward sweep and its adjoint) against the recurrence written as a torch loop + autograd."""
    from medical_image_classification_amd.cnn_mamba import _ChunkCarry
    b, c, g, n, hg, p = cfg
    gen = torch.Generator().manual_seed(c * 7 + p)
    S = torch.randn(b, c, g, n, hg, p, generator=gen)
    d = torch.rand(b, c, g * hg, generator=gen) * 0.9 + 0.05
    go = torch.randn(b, c, g, n, hg, p, generator=gen)
    Sr, dr = S.clone().double().requires_grad_(), d.clone().double().requires_grad_()
    carry, outs = torch.zeros(b, g, n, hg, p, dtype=torch.float64), []
    for z in range(c):
        outs.append(carry)
        carry = carry * dr[:, z].view(b, g, 1, hg, 1) + Sr[:, z]
    ref = torch.stack(outs, dim=1) + 0.0 * (Sr.sum() + dr.sum())      # c == 1: keep the graph connected
    ref.backward(go.double())
    Sd, dd = S.to(dev()).requires_grad_(), d.to(dev()).requires_grad_()
    out = _ChunkCarry.apply(Sd, dd)
    out.backward(go.to(dev()))
    close(out, ref, 1e-5, "S_in")
    close(Sd.grad, Sr.grad, 1e-5, "dS")
    close(dd.grad, dr.grad, 1e-4, "ddecay")


def test_chunked_ssd_carry_as_gemm_matches_kernel(monkeypatch):
    """The two evaluations of the chunk-state carry inside _ssd_chunked (HIP sweep / decay-matrix GEMM, the latter also the
    path for headdim % 4 != 0) give the same operator."""
    from medical_image_classification_amd import cnn_mamba as cm
    gen = torch.Generator().manual_seed(3)
    b, l, h, p, g, n = 2, 200, 4, 8, 2, 24
    mk = lambda *s: torch.randn(*s, generator=gen).to(dev())
    x, dt, B, C = mk(b, l, h, p), mk(b, l, h), mk(b, l, g, n), mk(b, l, g, n)
    A = -(torch.rand(h, generator=gen) * 2 + 0.1).to(dev())
    res = []
    for flag in (True, False):
        monkeypatch.setattr(cm, "SSD_CARRY_KERNEL", flag)
        ins = [t.clone().requires_grad_() for t in (x, dt, A, B, C)]
        y = cm._SSDChunked.apply(*ins, None, None, True)
        y.square().sum().backward()
        res.append([y.detach()] + [t.grad for t in ins])
    for a, r, name in zip(res[0], res[1], ("y", "dx", "ddt", "dA", "dB", "dC")):
        close(a, r, 1e-4, name)


@pytest.mark.parametrize("bf16", [False, True])
def test_stem_batchnorm_matches_nn_batchnorm(bf16):
    """ConvTConvPW._bn (reductions + affine pass) == nn.BatchNorm2d in training mode: output, input/weight/bias gradients,
    running statistics and the batch counter."""
    from medical_image_classification_amd.cnn_mamba import ConvTConvPW
    torch.manual_seed(2)
    stem = ConvTConvPW(in_channels=3).to(dev()).train()
    ref = torch.nn.BatchNorm2d(3).to(dev()).train()
    with torch.no_grad():
        stem.bn.weight.uniform_(0.5, 1.5); stem.bn.bias.uniform_(-0.5, 0.5)
        ref.weight.copy_(stem.bn.weight); ref.bias.copy_(stem.bn.bias)
    x = torch.randn(4, 3, 40, 36, device=dev()) * 2 + 0.5
    if bf16:
        x = x.bfloat16()
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    g = torch.randn_like(x)
    ya, yb = stem._bn(xa), ref(xb)
    assert ya.dtype == yb.dtype
    ya.backward(g); yb.backward(g)
    tol = 2e-2 if bf16 else 1e-5
    close(ya, yb, tol, "y"); close(xa.grad, xb.grad, tol, "dx")
    close(stem.bn.weight.grad, ref.weight.grad, 2e-3 if bf16 else 1e-4, "dweight"); close(stem.bn.bias.grad, ref.bias.grad, 2e-3 if bf16 else 1e-4, "dbias")
    close(stem.bn.running_mean, ref.running_mean, 1e-5, "running_mean"); close(stem.bn.running_var, ref.running_var, 1e-5, "running_var")
    assert int(stem.bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    stem.eval()
    assert torch.equal(stem._bn(x), stem.bn(x))                          # eval mode: the module itself


@pytest.mark.parametrize("cfg", [(2, 5, 7, 96), (1, 3, 3, 1024), (3, 2, 5, 70), (1, 1, 1, 130), (2, 9, 4, 256)])
@pytest.mark.parametrize("bf16", [False, True])
def test_rms_gate_kernels_vs_restatement(cfg, bf16):
    """RMSNormGated(norm_before_gate=False) through ms_rms_gate_fwd / _bwd against the float64 restatement: output and the
    gradients of x, z (a strided channel slice of a wider tensor, as in the modules) and weight."""
    from medical_image_classification_amd.cnn_mamba import RMSNormGated
    B, H, W, D = cfg
    gen = torch.Generator().manual_seed(D + H)
    x = torch.randn(B, H, W, D, generator=gen)
    wide = torch.randn(B, H, W, D + 24, generator=gen)
    g = torch.randn(B, H, W, D, generator=gen)
    wt = torch.randn(D, generator=gen)
    if bf16:
        wide, g = wide.bfloat16().float(), g.bfloat16().float()
    xr, zr, wr = x.double().requires_grad_(), wide[..., 8:8 + D].double().requires_grad_(), wt.double().requires_grad_()
    ref = ssd_oracle.rmsnorm_gated_ref(xr, zr, wr, 1e-5, False)
    ref.backward(g.double())
    n = RMSNormGated(D, eps=1e-5, norm_before_gate=False, group_size=D).to(dev())
    with torch.no_grad():
        n.weight.copy_(wt)
    dt = torch.bfloat16 if bf16 else torch.float32
    xd = x.to(dev()).requires_grad_()
    wided = wide.to(dev(), dt).requires_grad_()
    out = n(xd, wided[..., 8:8 + D], out_dtype=dt)
    assert out.dtype == dt
    out.backward(g.to(dev(), dt))
    tol = 2e-2 if bf16 else 1e-4
    close(out, ref, tol, "out")
    close(xd.grad, xr.grad, tol, "dx")
    close(wided.grad[..., 8:8 + D], zr.grad, tol, "dz")
    assert float(wided.grad[..., :8].abs().max()) == 0.0
    close(n.weight.grad, wr.grad, 1e-3 if not bf16 else 1e-2, "dweight")


@pytest.mark.parametrize("cfg", [(2, 5, 7, (6, 3, 3, 2)), (1, 8, 8, (64, 16, 16, 1)), (3, 1, 4, (5,)), (1, 9, 2, (4, 4))])
def test_channel_last_cross_scan_merge_kernels(cfg):
    """ms_cross_scan_nhwc / ms_cross_merge_nhwc against the index tables of the reference orders (_scan_orders =
    CNN_Mamba.py:494-498): bit-exact gather, merge in the reference's add order, each the other's adjoint."""
    from medical_image_classification_amd.cnn_mamba import _PixelsFromSeq, _SeqFromPixels, _scan_orders
    B, H, W, widths = cfg
    conv, L = sum(widths), H * W
    gen = torch.Generator().manual_seed(conv + L)
    xc = torch.randn(B, H, W, conv, generator=gen).to(dev()).requires_grad_()
    idx, inv = _scan_orders(H, W, dev())
    outs = _SeqFromPixels.apply(xc, widths)
    col = 0
    for w, o in zip(widths, outs):
        want = xc.detach().reshape(B, L, conv)[:, idx.t().reshape(-1), col:col + w].view(B, L, 4, w)
        assert torch.equal(o, want)
        col += w
    gs = [torch.randn(o.shape, generator=gen).to(dev()) for o in outs]
    torch.autograd.backward(outs, gs)
    col = 0
    for w, g in zip(widths, gs):
        want = ((g[:, inv[0], 0] + g[:, inv[2], 2]) + g[:, inv[1], 1]) + g[:, inv[3], 3]
        assert torch.equal(xc.grad.reshape(B, L, conv)[..., col:col + w], want)
        col += w
    y = torch.randn(B, L, 4, widths[0], generator=gen).to(dev()).requires_grad_()
    m = _PixelsFromSeq.apply(y, H, W)
    want = ((y[:, inv[0], 0] + y[:, inv[2], 2]) + y[:, inv[1], 1]) + y[:, inv[3], 3]
    assert torch.equal(m, want.detach())
    gm = torch.randn(m.shape, generator=gen).to(dev())
    m.backward(gm)
    assert torch.equal(y.grad, gm[:, idx.t().reshape(-1)].view(B, L, 4, -1))


@pytest.mark.parametrize("cfg", [(64, 64, 9, 7, 2), (128, 64, 14, 14, 3), (32, 16, 5, 12, 1), (64, 32, 40, 3, 2),
                                 (512, 64, 3, 5, 8)])          # the last one: enough waves for the 16-channel / 16-states-per-lane variant
def test_ssd_all_direction_forward_launch_matches_four_launches(cfg, monkeypatch):
    """MS_SCAN_BC_MAP(4): the 64-state SSD forward (4 directions x 16 states) in ONE launch == four one-direction launches that
    add up, and the slice-major saved states it writes drive the same backward: output and every gradient."""
    from medical_image_classification_amd import cnn_mamba as cm
    from medical_image_classification_amd import ss2d_fused as sf
    d_model, headdim, H, W, B = cfg
    torch.manual_seed(H * W)
    m = cm.SS2D_with_SSD(d_model=d_model, d_state=16, headdim=headdim).to(dev())
    with torch.no_grad():
        m.Ds.add_(torch.randn_like(m.Ds) * 0.3); m.A_logs.add_(torch.randn_like(m.A_logs) * 0.3); m.dt_bias.add_(torch.randn_like(m.dt_bias))
    monkeypatch.setattr(cm, "SSD_CHUNKED_MIN_STATE", 0)
    u = torch.randn(B, H, W, d_model, device=dev())
    g = torch.randn(B, H, W, d_model, device=dev())
    res = []
    for one in (True, False):
        monkeypatch.setattr(sf, "SSD_ONE_LAUNCH_FWD", one)
        m.zero_grad(set_to_none=True)
        ui = u.clone().requires_grad_()
        y = m(ui)
        y.backward(g)
        res.append((y.detach(), ui.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    (y1, du1, p1), (y0, du0, p0) = res
    close(y1, y0, 1e-5, "y")
    close(du1, du0, 1e-4, "du")
    for k in p0:
        close(p1[k], p0[k], 1e-4, k)


def _fusion_224_inputs():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fusion_loss_224.npz"))
    torch.manual_seed(int(g["seed"]))                         # the reference-run script's generator (tools/make_golden_fusion.py)
    vis, ir = torch.rand(1, 3, 224, 224), torch.rand(1, 3, 224, 224)
    assert abs(vis.double().sum().item() - float(g["vis_sum"])) < 1e-6 and abs(ir.double().sum().item() - float(g["ir_sum"])) < 1e-6
    return g, vis, ir


def test_fusion_loss_224_on_gpu_matches_reference_vectors():
    """FusionLoss forward + backward ON THE GPU at 224 x 224 (BASELINE.json configs[4]; the size at which MIOpen's backward of
    the reference's single-channel 11x11 conv2d chain faulted) against vectors from running the reference's
    loss.py on CPU (tests/golden/fusion_loss_224.npz): every loss term and d total / d fused image."""
    from medical_image_classification_amd.fusion_loss import FusionLoss
    g, vis, ir = _fusion_224_inputs()
    gen = torch.from_numpy(g["gen"]).to(dev()).requires_grad_()
    total, loss_in, ssim_value, loss_grad = FusionLoss().to(dev())(vis.to(dev()), ir.to(dev()), gen)
    total.backward()
    torch.cuda.synchronize()
    for name, v in (("total", total), ("loss_in", loss_in), ("ssim_value", ssim_value), ("loss_grad", loss_grad)):
        np.testing.assert_allclose(v.detach().cpu().numpy(), g[name], rtol=1e-4, atol=1e-5, err_msg=name)
    ref = g["dgen"]
    np.testing.assert_allclose(gen.grad.cpu().numpy(), ref, rtol=1e-3, atol=2e-4 * float(np.abs(ref).max()))


def test_vfefm_fusion_step_at_224():
    """One bf16-autocast fusion_step of the model CrossMamba/train.py:80-91 builds at BASELINE.json configs[4]'s image size
    (2 x 3 x 224 x 224 pairs, batch 1): finite loss terms, a finite gradient on every used parameter, weights move."""
    from medical_image_classification_amd.train_fusion import build_fusion_model, fusion_step, synthetic_pair
    from medical_image_classification_amd.fusion_loss import FusionLoss
    from medical_image_classification_amd.train import make_adam
    torch.manual_seed(0)
    net = build_fusion_model().to(dev()).train()
    opt = make_adam(net.parameters(), lr=2e-4)
    vis, ir = synthetic_pair(1, 224, dev())
    before = net.final_conv.weight.detach().clone()
    terms = fusion_step(net, opt, FusionLoss().to(dev()), vis, ir, torch.bfloat16)
    torch.cuda.synchronize()
    assert all(torch.isfinite(t).item() for t in terms)
    used = 0
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all().item(), n
            used += 1
    assert used > 1000
    assert not torch.equal(before, net.final_conv.weight.detach())


def test_vfefm_fusion_step_at_224_batch_32():
    """BASELINE.json configs[4] as it is benchmarked: 32 pairs of 3 x 224 x 224, the model CrossMamba/train.py:80-91 builds, two
    bf16-autocast fusion steps (the second one runs with refreshed bf16 weight copies and a re-used arena): finite loss terms that move,
    a finite gradient on every used parameter, peak device memory below 100 GiB (the chunk states of the stage-0 SSD scans are
    recomputed in the backward above MEDSCAN_SSD_KEEP_STATE_GB)."""
    from medical_image_classification_amd.train_fusion import build_fusion_model, fusion_step, synthetic_pair
    from medical_image_classification_amd.fusion_loss import FusionLoss
    from medical_image_classification_amd.train import make_adam
    import gc
    torch.manual_seed(0)
    gc.collect(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()                                  # whatever earlier tests of this process still hold
    net = build_fusion_model().to(dev()).train()
    opt = make_adam(net.parameters(), lr=2e-4)
    crit = FusionLoss().to(dev())
    vis, ir = synthetic_pair(32, 224, dev())
    t1 = fusion_step(net, opt, crit, vis, ir, torch.bfloat16)
    t2 = fusion_step(net, opt, crit, vis, ir, torch.bfloat16)
    torch.cuda.synchronize()
    assert all(torch.isfinite(t).item() for t in t1 + t2)
    assert float(t2[0]) != float(t1[0])                                   # the weights moved between the two steps
    used = 0
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all().item(), n
            used += 1
    assert used > 1000
    assert torch.cuda.max_memory_allocated() - base < 100 * 2 ** 30, (torch.cuda.max_memory_allocated() - base) / 2 ** 30


@pytest.mark.parametrize("cfg", [(2, 14, 14, 64, 4, 16), (1, 9, 5, 32, 2, 16), (2, 28, 28, 96, 3, 32)])
def test_ssd_backward_all_slices_in_one_launch(cfg, monkeypatch):
    """ms_selective_scan_bwd with MS_SCAN_BC_MAP(4) (csrc/scan_bwd_ssd.hip: the four direction slices of the state axis in one
    launch) against one accumulating launch per slice: every gradient of the 4-direction SSD scan."""
    from medical_image_classification_amd import ss2d_fused
    B, H, W, Ds, nheads, headdim = cfg
    N = 16
    assert nheads * headdim == Ds
    gen = torch.Generator().manual_seed(21)
    conv = Ds + 2 * N + nheads
    xc0 = torch.randn(B, H, W, conv, generator=gen).cuda()
    As0 = -(torch.rand(4 * nheads, generator=gen) + 0.5).cuda()
    Dsv0 = torch.randn(4 * nheads, generator=gen).cuda()
    bias0 = (torch.rand(4 * nheads, generator=gen) - 3.0).cuda()
    gy = torch.randn(B, H * W, Ds, generator=gen).cuda()
    res = []
    for one in (True, False):
        monkeypatch.setattr(ss2d_fused, "SSD_ONE_LAUNCH_BWD", one)
        ts = [t.clone().requires_grad_() for t in (xc0, As0, Dsv0, bias0)]
        y = ss2d_fused._SSDScanMerge.apply(ts[0], ts[1], ts[2], ts[3], H, W, Ds, N, nheads, headdim, False)
        y.backward(gy)
        res.append((y.detach(), [t.grad for t in ts]))
    (ya, ga), (yb, gb) = res
    assert torch.equal(ya, yb)
    for a, b, name in zip(ga, gb, ("dxc", "dA", "dD", "dbias")):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(b.abs().max()), err_msg=name)


def test_vfefm_224_batch4_matches_cpu_oracle():
    """BASELINE.json configs[4]'s geometry -- two 3 x 224 x 224 modalities, batch 4 -- on a depth-reduced VFEFM (one block per
    stage, dims 64..512, d_state 16: every scan at its full-size sequence lengths 3136 / 784 / 196 / 49, which is what the toy 64 x 64
    test cannot exercise): fused image, loss and > 60 parameter gradients against the CPU oracle modules, whose SSD scans run on the
    pinned C S6 oracle by expansion (tests/test_ssd_cpu.py holds that to the float64 loop).  fp32 on both sides.  PARITY UNPINNED
    against the reference's Triton dependency."""
    from medical_image_classification_amd.crossmamba import VFEFM
    from medical_image_classification_amd.fusion_loss import FusionLoss
    cfg = dict(depths=[1, 1, 1, 1], dims=[64, 128, 256, 512], depths_decoder=[1, 1, 1, 1], dims_decoder=[512, 256, 128, 64],
               d_state=16, drop_path_rate=0.0)
    torch.manual_seed(12)
    net, ref = VFEFM(**cfg), VFEFM(**cfg)
    ref.load_state_dict(net.state_dict())
    ssd_oracle.install_ssd(ref, by_expansion=True)
    try:
        net.to(dev()).train(); ref.train()
        x1, x2 = torch.rand(4, 3, 224, 224), torch.rand(4, 3, 224, 224)
        out_r = ssd_oracle.vfefm_forward_oracle(ref, x1, x2)
        out_d = net(x1.to(dev()), x2.to(dev()))
        assert tuple(out_d.shape) == (4, 1, 224, 224)
        close(out_d, out_r, 3e-3, "fused image")
        crit = FusionLoss()
        lr_ = crit(x1, x2, out_r.clamp(0, 1))[0]
        ld_ = crit.to(dev())(x1.to(dev()), x2.to(dev()), out_d.clamp(0, 1))[0]
        lr_.backward(); ld_.backward()
        assert abs(float(ld_.detach()) - float(lr_.detach())) <= 1e-3 * max(1.0, abs(float(lr_.detach())))
        pr = dict(ref.named_parameters())
        checked = 0
        for k, p in net.named_parameters():
            if pr[k].grad is None:
                assert p.grad is None, k
                continue
            if ("fusion" in k or "self_attention" in k or "final" in k or "in_proj" in k) and "norm" not in k:
                a, r = p.grad.detach().cpu().double().flatten(), pr[k].grad.double().flatten()
                cos = float(a @ r / (a.norm() * r.norm()).clamp_min(1e-300))
                rel = float((a - r).norm() / r.norm().clamp_min(1e-300))
                assert cos >= 0.999 and rel <= 5e-2, (k, cos, rel)       # fp32 vs fp32, atomics-ordered sums over 4 x 3136 positions; 4-element vectors (A_logs) included
                checked += 1
        assert checked > 60
    finally:
        ssd_oracle.install_ssd(torch.nn.Module())          # back to the float64 loop for the other tests of this process


@pytest.mark.parametrize("cfg", [(2, 200, 4, 64, False), (1, 64, 2, 128, True), (2, 49, 8, 512, False), (1, 1, 3, 64, False), (2, 65, 2, 64, True),
                                 (1, 127, 1, 192, False)])
@pytest.mark.parametrize("recompute", [False, True])
def test_ssd_chunk_mfma_kernels_vs_restatement(cfg, recompute, monkeypatch):
    """csrc/ssd_chunk.hip (the chunked SSD evaluation on the exact-fp32 matrix instruction: forward AND backward, no torch.matmul) against
    the sequential float64 restatement: output and all seven gradients at 1e-3 / 2e-3 (measured ~1e-6) -- chunk-ragged lengths (1, 49,
    65, 127, 200), 64 / 128 / 192 / 512 states, D per head and per (head, channel); `recompute`: the entering states are not kept for the
    backward (MEDSCAN_SSD_KEEP_STATE_GB = 0) but recomputed.  And no library GEMM runs: aten mm / bmm / matmul are never dispatched."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from medical_image_classification_amd import cnn_mamba as cm
    b, l, h, n, hdim_D = cfg
    p = 64
    monkeypatch.setattr(cm, "SSD_CHUNKED_MIN_STATE", 64)
    if recompute:
        monkeypatch.setattr(cm, "SSD_KEEP_STATE_BYTES", 0)
    gen = torch.Generator().manual_seed(l + n)
    mk = lambda *s: torch.randn(*s, generator=gen)
    x, dt, B, C = mk(b, l, h, p), mk(b, l, h) - 1.0, mk(b, l, 1, n) * 0.3, mk(b, l, 1, n) * 0.3
    A = -torch.rand(h, generator=gen) * 4 - 0.2
    D = mk(h, p) if hdim_D else mk(h)
    bias = mk(h) * 0.5
    gy = mk(b, l, h, p)
    cpu = [t.clone().requires_grad_() for t in (x, dt, A, B, C, D, bias)]
    gpu = [t.to(dev()).requires_grad_() for t in (x, dt, A, B, C, D, bias)]
    yr = ssd_oracle.ssd_scan_ref(cpu[0], cpu[1], cpu[2], cpu[3], cpu[4], D=cpu[5], dt_bias=cpu[6], dt_softplus=True)
    yr.backward(gy)

    class Audit(TorchDispatchMode):
        def __init__(self):
            super().__init__(); self.seen = []
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            nm = func.__name__.split(".")[0]
            if nm in ("mm", "addmm", "bmm", "baddbmm", "matmul", "einsum"):
                self.seen.append(nm)
            return func(*args, **(kwargs or {}))

    with Audit() as a:
        yd = cm.mamba_chunk_scan_combined(gpu[0], gpu[1], gpu[2], gpu[3], gpu[4], chunk_size=256, D=gpu[5], z=None, dt_bias=gpu[6], dt_softplus=True)
        yd.backward(gy.to(dev()))
    assert not a.seen, a.seen
    close(yd, yr, 1e-3, "y")
    for name, g_, r_ in zip(("dx", "ddt", "dA", "dB", "dC", "dD", "dbias"), gpu, cpu):
        close(g_.grad, r_.grad, 2e-3, name)


def test_ssd_chunk_mfma_kernels_plain_dt_and_error_returns():
    """dt_softplus = False, no bias, no D through the kernels; shapes they do not take fall back (n = 40) or are refused at the C ABI."""
    from medical_image_classification_amd import _lib, cnn_mamba as cm
    gen = torch.Generator().manual_seed(9)
    x, dt = torch.randn(1, 70, 2, 64, generator=gen), torch.rand(1, 70, 2, generator=gen) * 0.5
    A, B, C = -torch.ones(2), torch.randn(1, 70, 1, 64, generator=gen) * 0.3, torch.randn(1, 70, 1, 64, generator=gen) * 0.3
    cpu = [t.clone().requires_grad_() for t in (x, dt, A, B, C)]
    gpu = [t.to(dev()).requires_grad_() for t in (x, dt, A, B, C)]
    yr = ssd_oracle.ssd_scan_ref(*cpu)
    yd = cm.mamba_chunk_scan_combined(*gpu, chunk_size=64)
    g = torch.randn(1, 70, 2, 64, generator=gen)
    yr.backward(g); yd.backward(g.to(dev()))
    close(yd, yr, 1e-3, "y")
    for name, g_, r_ in zip(("dx", "ddt", "dA", "dB", "dC"), gpu, cpu):
        close(g_.grad, r_.grad, 2e-3, name)
    assert not cm._ssd_kernels_ok(torch.zeros(1, 4, 2, 64, device=dev()), torch.zeros(1, 4, 1, 40, device=dev()))      # 40 states: not a tile multiple
    lib, st = _lib.lib(), _lib.current_stream_ptr(dev())
    z = torch.zeros(64 * 64 * 4, device=dev()); q = z.data_ptr()
    assert lib.ms_ssd_chunk_fwd(None, q, q, q, q, None, 1, q, q, q, q, q, q, 1, 64, 1, 64, 64, st) == -1
    assert lib.ms_ssd_chunk_fwd(q, q, q, q, q, None, 1, q, q, q, q, q, q, 1, 64, 1, 32, 64, st) == -2          # headdim 32
    assert lib.ms_ssd_chunk_fwd(q, q, q, q, q, None, 1, q, q, q, q, q, q, 1, 64, 1, 64, 40, st) == -2          # 40 states
    assert lib.ms_ssd_chunk_bwd_off(q, q, q, q, None, q, 1, 64, 1, 64, 64, st) == -1
