"""GPU tests of the launch-tail machinery (round 2): the per-pass zero arena, the one-launch bf16 weight copies, the conv
branch's shadow-weight convolution and its 1x1 convolution + bias + ReLU on the MFMA GEMM, the one-pass block-input gradient
(BlockFrame / ms_block_head_bwd), the ReLU mask folded into the block tail's backward, and PatchMerging's tap gather.
Each fused path is held to the plain torch formulation of the same arithmetic (MedMamba.py:196-200, 517-527, 531-538)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_bf16_weight_copies_one_launch_refresh_and_staleness():
    from medical_image_classification_amd import shadow
    torch.manual_seed(0)
    ps = [nn.Parameter(torch.randn(s, device=dev())) for s in ((48, 48, 3, 3), (192, 96), (96, 96, 1, 1), (7,), (384, 384, 3, 3))]
    conv = [True, False, True, False, True]

    def want(p, c):
        t = p.detach().to(torch.bfloat16)
        return t.contiguous(memory_format=torch.channels_last) if (c and p.dim() == 4) else t

    for p, c in zip(ps, conv):
        s = shadow.bf16(p, conv=c)
        assert s.dtype == torch.bfloat16
        assert s.is_contiguous(memory_format=torch.channels_last) if (c and p.dim() == 4) else s.is_contiguous()
        assert torch.equal(s, want(p, c))                              # same rounding as torch's cast, channels_last memory order
    # an in-place update (what an optimizer step is) makes every copy stale; ONE request refreshes them all
    with torch.no_grad():
        for p in ps:
            p.mul_(1.5).add_(0.25)
    first = shadow.bf16(ps[0], conv=True)
    reg = shadow._registry(dev().index)
    tab = reg.table
    assert tab is not None and tab[3] >= len(ps)                        # the refresh covered (at least) all five in one table (>= one piece each)
    for p, c in zip(ps, conv):
        sh = shadow._BY_ID[(id(p), c)]
        assert not shadow._stale(p, sh, reg.epoch)
        assert torch.equal(sh.t, want(p, c))
    assert first.data_ptr() == shadow.bf16(ps[0], conv=True).data_ptr()        # cached: same tensor
    # the input-gradient convolution's weight: taps flipped, in / out channels swapped, (Ci, kh, kw, Co) memory
    fl = shadow.bf16(ps[0], conv="flip")
    assert torch.equal(fl, ps[0].detach().to(torch.bfloat16).flip(2, 3).permute(1, 2, 3, 0).contiguous())
    # ragged tiles (64 x 16 channel tiles), other tap counts (tiled up to 9 taps, element-wise above: the 4x4 patch embedding)
    for shape in ((70, 24, 3, 3), (96, 3, 4, 4), (50, 20, 2, 2), (16, 130, 3, 3), (5000, 3)):
        q = nn.Parameter(torch.randn(shape, device=dev()))
        c = len(shape) == 4
        assert torch.equal(shadow.bf16(q, conv=c), want(q, c))
        if c:
            assert torch.equal(shadow.bf16(q, conv="flip"), q.detach().to(torch.bfloat16).flip(2, 3).permute(1, 2, 3, 0).contiguous())
    # writes through .data do not bump the version counter: invalidate() (VSSM.forward does it once per training step) covers them
    ps[1].data.add_(1.0)
    assert not torch.equal(shadow._BY_ID[(id(ps[1]), False)].t, want(ps[1], False))
    shadow.invalidate(dev())
    assert torch.equal(shadow.bf16(ps[1]), want(ps[1], False))
    # a fused optimizer step does NOT bump the version counters: a process-wide optimizer post-step hook invalidates the copies
    fo = torch.optim.Adam([ps[4]], lr=0.1, fused=True)
    ps[4].grad = torch.ones_like(ps[4])
    fo.step()
    assert torch.equal(shadow.bf16(ps[4], conv=True), want(ps[4], True))
    # ... and so does the backward of every Function that used one (here: a convolution through the cached weight)
    from medical_image_classification_amd.block_ops import _conv2d
    conv = nn.Conv2d(48, 48, 3, padding=1).to(dev())
    opt = torch.optim.Adam(conv.parameters(), lr=0.1, fused=True)
    # (6 x 6 map: below the direct kernel's 49-pixel threshold, so both sides run the same library kernel and compare bit for bit)
    x = torch.randn(2, 48, 6, 6, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(2):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = _conv2d(conv, x)
            assert torch.equal(y, F.conv2d(x, conv.weight, None, conv.stride, conv.padding))      # the cached copy is current
        y.float().square().mean().backward()
        opt.step()
        opt.zero_grad()


def test_conv_with_cached_weight_matches_autocast_conv():
    from medical_image_classification_amd.block_ops import _conv2d
    torch.manual_seed(1)
    conv = nn.Conv2d(48, 48, 3, padding=1).to(dev())
    x = torch.randn(4, 48, 14, 14, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.randn(4, 48, 14, 14, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya = _conv2d(conv, xa)
        (dxa, dwa) = torch.autograd.grad(ya, (xa, conv.weight), g)
        yb = F.conv2d(xb, conv.weight, None, conv.stride, conv.padding)
        (dxb, dwb) = torch.autograd.grad(yb, (xb, conv.weight), g)
    # forward and input gradient run on ms_conv3x3_nhwc_bf16 (fp32 accumulation in another order than MIOpen's, bf16 outputs)
    assert ya.dtype == torch.bfloat16 and ya.is_contiguous(memory_format=torch.channels_last)
    for got, want in ((ya, yb), (dxa, dxb)):
        np.testing.assert_allclose(got.float().detach().cpu().numpy(), want.float().detach().cpu().numpy(), rtol=2e-2,
                                   atol=1e-2 * float(want.float().abs().max()))
    assert dwa.dtype == torch.float32 and dwa.is_contiguous()
    np.testing.assert_allclose(dwa.cpu().numpy(), dwb.cpu().numpy(), rtol=1e-2, atol=2e-2 * float(dwb.abs().max()))


@pytest.mark.parametrize("cfg", [(2, 48, 56, 56), (3, 96, 28, 28), (2, 192, 14, 14), (1, 384, 7, 7), (2, 64, 9, 21), (1, 128, 16, 16), (1, 256, 8, 9), (1, 16, 1, 1), (2, 48, 17, 5)])
def test_direct_conv3x3_forward_and_input_gradient_vs_fp32(cfg):
    """ms_conv3x3_nhwc_bf16 (csrc/conv3x3.hip) and its input-gradient form (the same kernel on dy with the flipped / transposed
    weight copy) against fp32 torch on the bf16-rounded operands, at the conv-branch shapes and ragged ones (tile edges)."""
    from medical_image_classification_amd import shadow
    from medical_image_classification_amd.block_ops import _conv3x3_direct
    B, C, H, W = cfg
    torch.manual_seed(31)
    w = nn.Parameter(torch.randn(C, C, 3, 3, device=dev()) * (9 * C) ** -0.5)
    x = torch.randn(B, C, H, W, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, C, H, W, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = _conv3x3_direct(x, shadow.bf16(w, conv=True))
    dx = _conv3x3_direct(g, shadow.bf16(w, conv="flip"))
    xr = x.float().requires_grad_()
    yr = F.conv2d(xr, w.detach().to(torch.bfloat16).float(), padding=1)
    (dxr,) = torch.autograd.grad(yr, xr, g.float())
    for got, want, name in ((y, yr, "y"), (dx, dxr, "dx")):
        np.testing.assert_allclose(got.float().cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-2, atol=6e-3 * float(want.abs().max()),
                                   err_msg=name)
    # weight gradient: fp32 accumulation of bf16 products, written (no zero-fill), (Co, Ci, 3, 3) contiguous
    from medical_image_classification_amd.block_ops import _conv3x3_wgrad
    dw = _conv3x3_wgrad(x, g, w.shape)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_()
    (dwr,) = torch.autograd.grad(F.conv2d(x.float(), wr, padding=1), wr, g.float())
    assert dw.dtype == torch.float32 and dw.shape == w.shape and dw.is_contiguous()
    np.testing.assert_allclose(dw.cpu().numpy(), dwr.cpu().numpy(), rtol=2e-3, atol=2e-3 * float(dwr.abs().max()), err_msg="dW")


@pytest.mark.parametrize("cfg", [(2, 80, 48, 13, 19), (1, 32, 112, 9, 33), (3, 16, 64, 8, 16), (1, 144, 96, 20, 7), (2, 48, 80, 1, 40)])
def test_direct_conv3x3_with_unequal_channel_counts(cfg):
    """The same three kernels with Ci != Co -- partial output-channel blocks (Co not a multiple of 48), 16-channel tail slices behind full
    ones (Ci = 80, 144), one-slice inputs (Ci = 16, 32), ragged tiles: the model only has square convolutions, the entry points do not."""
    from medical_image_classification_amd.block_ops import _conv3x3_direct, _conv3x3_wgrad
    B, Ci, Co, H, W = cfg
    torch.manual_seed(37)
    w = (torch.randn(Co, Ci, 3, 3, device=dev()) * (9 * Ci) ** -0.5).to(torch.bfloat16)
    x = torch.randn(B, Ci, H, W, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, Co, H, W, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = _conv3x3_direct(x, w.contiguous(memory_format=torch.channels_last))                       # (Co, 3, 3, Ci) memory
    dx = _conv3x3_direct(g, w.flip(2, 3).permute(1, 2, 3, 0).contiguous())                        # (Ci, 3, 3, Co) memory, taps flipped
    assert y.shape == (B, Co, H, W) and dx.shape == (B, Ci, H, W)
    xr, wr = x.float().requires_grad_(), w.float().requires_grad_()
    yr = F.conv2d(xr, wr, padding=1)
    dxr, dwr = torch.autograd.grad(yr, (xr, wr), g.float())
    for got, want, name in ((y, yr, "y"), (dx, dxr, "dx")):
        np.testing.assert_allclose(got.float().cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-2, atol=6e-3 * float(want.abs().max()),
                                   err_msg=name)
    dw = _conv3x3_wgrad(x, g, w.shape)
    np.testing.assert_allclose(dw.cpu().numpy(), dwr.cpu().numpy(), rtol=2e-3, atol=2e-3 * float(dwr.abs().max()), err_msg="dW")


@pytest.mark.parametrize("cfg", [(2, 48, 56, 56), (3, 96, 9, 7), (64, 384, 7, 7)])
@pytest.mark.parametrize("premasked", [False, True])
def test_conv1x1_relu_on_the_mfma_gemm(cfg, premasked):
    """relu(conv1x1(x) + b): forward, dx, dW, db against fp32 torch on the bf16-rounded operands (the kernel rounds the fp32
    weight to bf16 while staging; accumulation is fp32).  premasked: the caller supplies dy * [y > 0] (block_tail does)."""
    from medical_image_classification_amd.block_ops import conv1x1_relu
    B, C, H, W = cfg
    torch.manual_seed(2)
    conv = nn.Conv2d(C, C, 1).to(dev())
    x = torch.randn(B, C, H, W, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    g = torch.randn(B, C, H, W, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv1x1_relu(conv, x, premasked=premasked)
    assert y is not None and y.dtype == torch.bfloat16 and y.shape == (B, C, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    gin = g * (y > 0) if premasked else g
    dx, dw, db = torch.autograd.grad(y, (x, conv.weight, conv.bias), gin)
    xr = x.detach().float().requires_grad_()
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_()
    br = conv.bias.detach().clone().requires_grad_()
    yr = F.relu(F.conv2d(xr, wr, br))
    dxr, dwr, dbr = torch.autograd.grad(yr, (xr, wr, br), g.float())
    tol = lambda t: 1e-2 * float(t.abs().max())
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), yr.detach().cpu().numpy(), rtol=1e-2, atol=tol(yr))
    # the mask is taken from the bf16 output: identical to the fp32 one except where |pre-activation| is below bf16 resolution
    np.testing.assert_allclose(dx.float().cpu().numpy(), dxr.cpu().numpy(), rtol=2e-2, atol=2 * tol(dxr))
    np.testing.assert_allclose(dw.cpu().numpy(), dwr.cpu().numpy(), rtol=2e-2, atol=2 * tol(dwr))
    np.testing.assert_allclose(db.cpu().numpy(), dbr.cpu().numpy(), rtol=2e-2, atol=2 * tol(dbr))
    assert dw.dtype == torch.float32 and dw.shape == conv.weight.shape


@pytest.mark.parametrize("cfg", [(64, 129, 40, True, True), (300, 68, 96, False, True), (257, 48, 48, True, False), (64, 192, 24, False, False)])
def test_gemm_bias_relu_epilogue(cfg):
    from medical_image_classification_amd.gemm_ops import gemm
    M, N, K, bf16_out, relu = cfg
    if bf16_out and N % 8 != 0:
        N = (N + 7) // 8 * 8
    torch.manual_seed(3)
    a = torch.randn(M, K, device=dev()).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev())
    b = torch.randn(N, device=dev())
    y = gemm(a, w, out_dtype=torch.bfloat16 if bf16_out else torch.float32, bias=b, relu=relu)
    ref = a.float() @ w.to(torch.bfloat16).float().t() + b
    if relu:
        ref = F.relu(ref)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2 if bf16_out else 1e-4,
                               atol=(1e-2 if bf16_out else 1e-4) * float(ref.abs().max()))
    with pytest.raises(RuntimeError, match="store modes"):
        gemm(a, w, bias=b, accumulate=True)


@pytest.mark.parametrize("dts", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32), (torch.float32, torch.bfloat16)])
@pytest.mark.parametrize("shape", [(2, 56, 56, 96), (3, 5, 7, 16), (1, 1, 1, 8)])
def test_block_frame_gives_the_input_gradient_in_one_pass(shape, dts):
    """split_halves + block_tail sharing a BlockFrame == the same graph through autograd's concat + accumulation add, bit for bit
    (two fp32 terms per element: the add is commutative)."""
    from medical_image_classification_amd.block_ops import BlockFrame, block_tail, split_halves
    B, H, W, C = shape
    torch.manual_seed(4)
    x0 = torch.randn(shape, device=dev())
    g = torch.randn(shape, device=dev())
    scale = torch.rand(B, device=dev()) + 0.5
    grads = []
    for use_frame in (False, True):
        x = x0.clone().requires_grad_()
        frame = BlockFrame() if use_frame else None
        left, right = split_halves(x, frame)
        l2 = (left * 1.5).to(dts[0])
        r2 = torch.tanh(right).to(dts[1])
        out = block_tail(l2, r2, x, scale, frame=frame)
        out.backward(g)
        assert frame is None or frame.dout is None          # consumed
        grads.append(x.grad.clone())
    assert torch.equal(grads[0], grads[1])


def test_block_tail_applies_the_relu_mask_of_its_left_input():
    from medical_image_classification_amd.block_ops import block_tail
    torch.manual_seed(5)
    B, H, W, C = 2, 9, 11, 32
    z0 = torch.randn(B, H, W, C // 2, device=dev())
    xr = torch.randn(B, H, W, C // 2, device=dev())
    inp = torch.randn(B, H, W, C, device=dev())
    g = torch.randn(B, H, W, C, device=dev())
    for dt in (torch.float32, torch.bfloat16):
        z = z0.to(dt).requires_grad_()
        out = block_tail(F.relu(z), xr, inp)
        (want,) = torch.autograd.grad(out, z, g)
        # the fused form: the producer of `left` passes its incoming gradient through unchanged
        class _Relu(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                return F.relu(t)

            @staticmethod
            def backward(ctx, d):
                return d
        z2 = z0.to(dt).requires_grad_()
        out2 = block_tail(_Relu.apply(z2), xr, inp, left_relu=True)
        (got,) = torch.autograd.grad(out2, z2, g)
        assert torch.equal(out, out2) and torch.equal(got, want)


@pytest.mark.parametrize("shape", [(2, 8, 6, 5), (64, 56, 56, 96)])
def test_patch_merging_tap_gather_bit_exact(shape):
    from medical_image_classification_amd.medmamba import _GatherTaps
    torch.manual_seed(6)
    x = torch.randn(shape, device=dev(), requires_grad=True)
    ref = torch.cat([x[:, i::2, j::2, :] for (i, j) in ((0, 0), (1, 0), (0, 1), (1, 1))], dim=-1)      # MedMamba.py:196-200
    y = _GatherTaps.apply(x)
    assert torch.equal(y, ref)
    g = torch.randn_like(ref)
    (gr,) = torch.autograd.grad(ref, x, g)
    (gy,) = torch.autograd.grad(y, x, g)
    assert torch.equal(gr, gy)


def _block_and_input(dim=96, hw=14, batch=4):
    from medical_image_classification_amd.medmamba import SS_Conv_SSM
    torch.manual_seed(7)
    blk = SS_Conv_SSM(hidden_dim=dim, drop_path=0.0, d_state=16).to(dev()).train()
    x = torch.randn(batch, hw, hw, dim, device=dev())
    return blk, x


def test_grad_arena_passes_do_not_alias(monkeypatch):
    """Gradients of consecutive backward passes come from different arena buffers: accumulating over two passes without
    zero_grad gives the sum, and a pass's gradients are not disturbed by the next pass."""
    from medical_image_classification_amd import arena
    blk, x = _block_and_input()
    params = [p for p in blk.parameters() if p.requires_grad]

    def run(xin):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(xin.clone().requires_grad_())
        y.float().square().mean().backward()

    def grads():
        return [p.grad.clone() if p.grad is not None else None for p in params]

    def zero():
        for p in params:
            p.grad = None

    monkeypatch.setattr(arena, "_ENABLED", False)
    zero(); run(x); ga = grads()
    zero(); run(2 * x); gb = grads()
    monkeypatch.setattr(arena, "_ENABLED", True)
    arena._ARENAS.clear()
    zero(); run(x)                                  # pass 1 learns the demand (plain torch.zeros)
    zero(); run(x); g1 = [p.grad for p in params]   # pass 2 runs out of the arena; keep the very tensors autograd stored
    a = arena._ARENAS[dev().index]
    assert a.last > 0
    snap = [t.clone() if t is not None else None for t in g1]
    run(2 * x)                                      # pass 3 accumulates INTO pass 2's gradients
    for p, s, u, v in zip(params, snap, ga, gb):
        if s is None:
            continue
        # atomics make the reductions order-dependent in the last bits: compare with a tolerance
        tol = 1e-3 * max(1e-6, float(u.abs().max()))
        np.testing.assert_allclose(s.float().cpu().numpy(), u.float().cpu().numpy(), rtol=2e-2, atol=10 * tol)
        np.testing.assert_allclose(p.grad.float().cpu().numpy(), (u + v).float().cpu().numpy(), rtol=2e-2,
                                   atol=10 * 1e-3 * max(1e-6, float((u + v).abs().max())))


def test_batchnorm_backward_writes_dx_in_the_input_dtype():
    """x fp32 (the block input's left half), dy bf16 (from the convolution behind it): dx comes out fp32 without a cast pass."""
    from medical_image_classification_amd.block_ops import batchnorm_relu
    torch.manual_seed(8)
    bn = nn.BatchNorm2d(48).to(dev()).train()
    ref = nn.BatchNorm2d(48).to(dev()).train()
    x0 = torch.randn(8, 14, 14, 96, device=dev())
    x = x0.clone().requires_grad_()
    left = x[..., :48].permute(0, 3, 1, 2)                           # strided NCHW view of the left half
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = batchnorm_relu(bn, left, False)
    g = torch.randn_like(y)
    assert y.dtype == torch.bfloat16
    y.backward(g)
    xr = x0.clone().requires_grad_()
    yr = ref(xr[..., :48].permute(0, 3, 1, 2).contiguous())
    yr.backward(g.float())
    assert x.grad.dtype == torch.float32
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-3, atol=1e-3 * float(xr.grad.abs().max()))


@pytest.mark.parametrize("cfg", [(2, 56, 56, 96, 3), (2, 14, 14, 384, 12), (1, 9, 7, 48, 5)])
def test_activation_inside_the_dt_projection_is_bit_identical(cfg, monkeypatch):
    """ms_dtproj_fwd_act + MS_SCAN_DELTA_ACTIVATED (delta' = softplus(delta + bias) formed by the bandwidth-bound projection kernel,
    skipped by the issue-bound scan kernels) against the scan applying it itself: the same function on the same values, so the
    outputs and every gradient agree bit for bit (up to the order of the atomics in the reductions)."""
    from medical_image_classification_amd import ss2d_fused
    from medical_image_classification_amd.medmamba import SS2D
    B, H, W, D, R = cfg
    torch.manual_seed(9)
    mod = SS2D(d_model=D // 2, d_state=16, dt_rank=R).to(dev()).train()
    x0 = torch.randn(B, H, W, D // 2, device=dev())
    g = torch.randn(B, H, W, D // 2, device=dev())
    res = []
    for act in (True, False):
        monkeypatch.setattr(ss2d_fused, "_DT_ACT", act)
        for p in mod.parameters():
            p.grad = None
        x = x0.clone().requires_grad_()
        y = mod(x)
        y.backward(g)
        res.append((y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in mod.named_parameters()}))
    (ya, dxa, ga), (yb, dxb, gb) = res
    assert torch.equal(ya, yb)
    np.testing.assert_allclose(dxa.cpu().numpy(), dxb.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(dxb.abs().max()))
    for n in ga:
        np.testing.assert_allclose(ga[n].cpu().numpy(), gb[n].cpu().numpy(), rtol=1e-4, atol=1e-5 * float(gb[n].abs().max()) + 1e-12, err_msg=n)


def test_ms_adam_matches_torch_adam_and_shares_its_state_dict():
    """adam.MsAdam (ms_adam_multi: every parameter in one launch) against torch.optim.Adam's single-tensor implementation over
    several steps -- tensors of 1 .. 3 chunks + tails, an UNALIGNED view (the data-parallel wrapper's flat-buffer gradients), a
    parameter that never gets a gradient; then the state_dict of one continues in the other (the reference's checkpoints hold
    `optimizer.state_dict()`, train.py / ddp_train.py)."""
    from medical_image_classification_amd.adam import MsAdam
    torch.manual_seed(5)
    shapes = [(1,), (3, 5), (4096,), (4097,), (3, 4099), (64, 3, 3, 3), (2, 8193)]
    base = [torch.randn(s, device=dev()) for s in shapes]
    pa = [nn.Parameter(b.clone()) for b in base] + [nn.Parameter(torch.ones(7, device=dev()))]      # the last one: never used
    pb = [nn.Parameter(b.clone()) for b in base] + [nn.Parameter(torch.ones(7, device=dev()))]
    oa, ob = MsAdam(pa, lr=1e-2), torch.optim.Adam(pb, lr=1e-2, foreach=False, fused=False)
    flat = torch.zeros(sum(b.numel() for b in base) + 1, device=dev())

    def set_grads(step):
        g = torch.Generator(device=dev()).manual_seed(100 + step)
        off = 1                                             # views at odd float offsets: not 16-byte aligned
        for a, b in zip(pa[:-1], pb[:-1]):
            gr = torch.randn(a.shape, device=dev(), generator=g) * (10.0 ** (step % 3 - 1))
            v = flat[off:off + a.numel()].view(a.shape); off += a.numel()
            v.copy_(gr)
            a.grad = v if step % 2 else gr.clone()
            b.grad = gr.clone()

    for step in range(5):
        set_grads(step)
        oa.step(); ob.step()
        for a, b in zip(pa, pb):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    assert pa[-1].grad is None and pa[-1] not in oa.state and torch.equal(pa[-1].detach(), torch.ones(7, device=dev()))
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert sa["state"][k].keys() == sb["state"][k].keys() and float(sa["state"][k]["step"]) == 5.0
        np.testing.assert_allclose(sa["state"][k]["exp_avg_sq"].cpu().numpy(), sb["state"][k]["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=1e-12)
    # cross-load: torch's state continues in MsAdam and vice versa
    oa2, ob2 = MsAdam(pa, lr=1e-2), torch.optim.Adam(pb, lr=1e-2, foreach=False, fused=False)
    oa2.load_state_dict(sb); ob2.load_state_dict(sa)
    set_grads(5)
    oa2.step(); ob2.step()
    for a, b in zip(pa, pb):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=4e-6, atol=2e-7)
    assert float(oa2.state_dict()["state"][0]["step"]) == 6.0
    # a group the kernel does not serve (weight decay) runs torch's implementation
    oc = MsAdam([nn.Parameter(torch.ones(5, device=dev()))], lr=1e-2, weight_decay=0.1)
    oc.param_groups[0]["params"][0].grad = torch.ones(5, device=dev())
    oc.step()
    assert float(oc.state_dict()["state"][0]["step"]) == 1.0


def test_ms_adam_step_with_a_closure_matches_torch_adam():
    """`optimizer.step(closure)`: torch.optim.Adam evaluates the closure FIRST and updates with the gradients it produced; a closure
    that does zero_grad(set_to_none=True) + backward creates NEW gradient tensors, so an implementation that collected the gradient
    lists before calling it would update with the previous step's gradients (or, on the very first step, not at all)."""
    from medical_image_classification_amd.adam import MsAdam
    torch.manual_seed(11)
    w0 = torch.randn(300, 17, device=dev()); x = torch.randn(64, 17, device=dev()); y = torch.randn(64, 300, device=dev())
    pa, pb = nn.Parameter(w0.clone()), nn.Parameter(w0.clone())
    oa, ob = MsAdam([pa], lr=1e-2), torch.optim.Adam([pb], lr=1e-2, foreach=False, fused=False)

    def closure_for(p, opt, it):
        def closure():
            opt.zero_grad(set_to_none=True)
            loss = ((x * (1.0 + it)) @ p.t() - y).square().mean()
            loss.backward()
            return loss
        return closure

    for it in range(4):
        la = oa.step(closure_for(pa, oa, it)); lb = ob.step(closure_for(pb, ob, it))
        assert la is not None and abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=4e-6, atol=2e-7)
    assert oa._ms_tables, "the one-launch path served the closure steps"
    assert float(oa.state_dict()["state"][0]["step"]) == 4.0


def test_ms_adam_resume_with_device_mapped_checkpoint_stays_on_the_one_launch_path(tmp_path):
    """ddp_train.py:142-150 (`--resume`): the checkpoint is loaded with `map_location=device`, which puts every `state['step']` on the
    GPU; MsAdam.load_state_dict brings the counters back to the host so that the step after a resume is still ms_adam_multi (and equals
    torch's Adam continued from the same state)."""
    from medical_image_classification_amd.adam import MsAdam
    torch.manual_seed(12)
    base = [torch.randn(s, device=dev()) for s in [(33,), (5000,), (16, 3, 3, 3)]]
    pa = [nn.Parameter(b.clone()) for b in base]; pb = [nn.Parameter(b.clone()) for b in base]
    oa, ob = MsAdam(pa, lr=1e-2), torch.optim.Adam(pb, lr=1e-2, foreach=False, fused=False)

    def grads(step):
        g = torch.Generator(device=dev()).manual_seed(step)
        for a, b in zip(pa, pb):
            a.grad = torch.randn(a.shape, device=dev(), generator=g); b.grad = a.grad.clone()

    for s in range(2):
        grads(s); oa.step(); ob.step()
    path = tmp_path / "ck.pth"
    torch.save({"optimizer": oa.state_dict(), "model": [p.detach() for p in pa]}, path)
    ck = torch.load(path, map_location=dev(), weights_only=True)
    assert all(st["step"].is_cuda for st in ck["optimizer"]["state"].values())        # the situation the fix is for
    oa2 = MsAdam(pa, lr=1e-2)
    oa2.load_state_dict(ck["optimizer"])
    assert all(st["step"].device.type == "cpu" for st in oa2.state.values())
    grads(2); oa2.step(); ob.step()
    assert oa2._ms_tables, "after a resume the update still runs as one ms_adam_multi launch"
    for a, b in zip(pa, pb):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=4e-6, atol=2e-7)
    assert float(oa2.state_dict()["state"][0]["step"]) == 3.0


@pytest.mark.parametrize("cfg", [(2, 3, 224, 224, 96), (3, 3, 32, 20, 48), (1, 4, 8, 8, 64)])
def test_patch_embedding_as_im2col_gemm(cfg):
    """PatchEmbed2D's 4 x 4 / stride 4 convolution under bf16 autocast = ms_patchify4_bf16 + ms_gemm_bf16_bias_act, its weight / bias
    gradient ms_gemm_bf16_wgrad_bias: against F.conv2d in fp32 on the bf16-rounded operands (output, dW, db); the im2col itself
    bit-exact against the view / permute formulation."""
    from medical_image_classification_amd import _lib
    from medical_image_classification_amd.medmamba import PatchEmbed2D, _patch_embed_gemm_ok
    B, C, H, W, E = cfg
    torch.manual_seed(E)
    pe = PatchEmbed2D(patch_size=4, in_chans=C, embed_dim=E, norm_layer=None).to(dev())
    x = torch.randn(B, C, H, W, device=dev())
    patches = torch.empty(B * (H // 4) * (W // 4), C * 16, device=dev(), dtype=torch.bfloat16)
    _lib.check(_lib.lib().ms_patchify4_bf16(x.data_ptr(), patches.data_ptr(), B, C, H, W, _lib.current_stream_ptr(dev())), "patchify")
    want = x.view(B, C, H // 4, 4, W // 4, 4).permute(0, 2, 4, 1, 3, 5).reshape(-1, C * 16).to(torch.bfloat16)
    assert torch.equal(patches, want)
    g = torch.randn(B, H // 4, W // 4, E, device=dev())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert _patch_embed_gemm_ok(pe.proj, x)
        y = pe(x)
    assert y.dtype == torch.float32 and y.shape == (B, H // 4, W // 4, E)
    dw, db = torch.autograd.grad(y, (pe.proj.weight, pe.proj.bias), g)
    wr = pe.proj.weight.detach().to(torch.bfloat16).float().requires_grad_()
    br = pe.proj.bias.detach().clone().requires_grad_()
    yr = F.conv2d(x.to(torch.bfloat16).float(), wr, br, stride=4).permute(0, 2, 3, 1)
    dwr, dbr = torch.autograd.grad(yr, (wr, br), g.to(torch.bfloat16).float())
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=1e-4, atol=1e-4 * float(yr.detach().abs().max()))
    np.testing.assert_allclose(dw.cpu().numpy(), dwr.cpu().numpy(), rtol=1e-3, atol=1e-3 * float(dwr.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), dbr.cpu().numpy(), rtol=1e-3, atol=1e-3 * float(dbr.abs().max()))
    # an input that needs a gradient keeps the convolution
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not _patch_embed_gemm_ok(pe.proj, x.clone().requires_grad_())


@pytest.mark.parametrize("cfg", [(2, 56, 56, 96), (3, 6, 10, 12), (1, 14, 14, 192), (2, 4, 2, 256), (2, 14, 14, 384), (1, 6, 4, 512)])   # last two: 1536 / 2048 channels (T / B stage 3)
@pytest.mark.parametrize("bf16", [False, True])
def test_patch_merging_gather_inside_the_layernorm(cfg, bf16):
    """ms_layernorm_taps_fwd/bwd (PatchMerging2D's 2 x 2 tap gather as an addressing mode of its LayerNorm, the scatter of the dx
    store) against F.layer_norm of the explicit concatenation in float64: output, dx, dgamma, dbeta."""
    from medical_image_classification_amd.block_ops import layernorm_taps, layernorm_taps_ok
    B, H, W, C = cfg
    torch.manual_seed(C)
    x = (torch.randn(B, H, W, C, device=dev()) * 2 + 0.3).requires_grad_()
    norm = nn.LayerNorm(4 * C).to(dev())
    with torch.no_grad():
        norm.weight.normal_(); norm.bias.normal_()
    assert layernorm_taps_ok(x, norm)
    y = layernorm_taps(x, norm, out_bf16=bf16)
    assert y.shape == (B, H // 2, W // 2, 4 * C) and y.dtype == (torch.bfloat16 if bf16 else torch.float32)
    g = torch.randn(B, H // 2, W // 2, 4 * C, device=dev())
    if bf16:
        g = g.to(torch.bfloat16)
    dx, dw, db = torch.autograd.grad(y, (x, norm.weight, norm.bias), g)
    xr = x.detach().double().requires_grad_()
    wr, br = norm.weight.detach().double().requires_grad_(), norm.bias.detach().double().requires_grad_()
    taps = torch.cat([xr[:, i::2, j::2, :] for (i, j) in ((0, 0), (1, 0), (0, 1), (1, 1))], dim=-1)
    yr = F.layer_norm(taps, (4 * C,), wr, br, norm.eps)
    dxr, dwr, dbr = torch.autograd.grad(yr, (xr, wr, br), g.double())
    tol = 2e-2 if bf16 else 1e-4
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), yr.detach().float().cpu().numpy(), rtol=tol, atol=tol * float(yr.detach().abs().max()))
    np.testing.assert_allclose(dx.cpu().numpy(), dxr.float().cpu().numpy(), rtol=1e-3, atol=1e-4 * float(dxr.abs().max()))
    np.testing.assert_allclose(dw.cpu().numpy(), dwr.float().cpu().numpy(), rtol=1e-3, atol=1e-4 * float(dwr.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), dbr.float().cpu().numpy(), rtol=1e-3, atol=1e-4 * float(dbr.abs().max()))
