"""GPU parity of the hand-written MFMA projection GEMM (csrc/gemm.hip) against float64 products of the SAME bf16-rounded
operands (so only the accumulation order differs): every operand layout / dtype the projections use, ragged edges in all
three dimensions, split-K accumulation, and the autograd wrapper against F.linear."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _round(t):
    return t.to(torch.bfloat16).double() if t.dtype == torch.float32 else t.double()


# (M, N, K): MedMamba-T projection shapes (batch reduced) + ragged ones
SHAPES = [(3136 * 2, 192, 48), (784 * 2, 152, 192), (196 * 2, 768, 192), (49 * 2, 384, 768), (200, 140, 96), (129, 68, 40),
          (1, 4, 8), (1000, 48, 96), (256, 224, 768)]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"{s[0]}x{s[1]}x{s[2]}" for s in SHAPES])
@pytest.mark.parametrize("a_f32", [False, True])
@pytest.mark.parametrize("out_bf16", [False, True])
def test_forward_and_input_grad_layouts(shape, a_f32, out_bf16):
    from medical_image_classification_amd.gemm_ops import gemm
    M, N, K = shape
    gen = torch.Generator(device=dev()).manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev(), generator=gen)
    if not a_f32:
        a = a.to(torch.bfloat16)
    w = torch.randn(N, K, device=dev(), generator=gen) * K ** -0.5
    od = torch.bfloat16 if out_bf16 else torch.float32
    y = gemm(a, w, out_dtype=od)                              # y = a W^T
    ref = _round(a) @ _round(w).t()
    tol = (8e-3 if out_bf16 else 2e-5) * float(ref.abs().max())
    np.testing.assert_allclose(y.double().cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=tol)
    dy = torch.randn(M, N, device=dev(), generator=gen)
    if not a_f32 and N % 8 == 0:                              # bf16 rows must be multiples of 16 bytes (x_proj's 4(R+2N) = 140.. columns are fp32)
        dy = dy.to(torch.bfloat16)
    dx = gemm(dy, w, b_trans=True, out_dtype=od)              # dx = dy W
    ref = _round(dy) @ _round(w)
    tol = (8e-3 if out_bf16 else 2e-5) * float(ref.abs().max())
    np.testing.assert_allclose(dx.double().cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=tol)


@pytest.mark.parametrize("shape", SHAPES, ids=[f"{s[0]}x{s[1]}x{s[2]}" for s in SHAPES])
@pytest.mark.parametrize("dt", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("splits", [1, 4])
def test_weight_grad_split_k(shape, dt, splits):
    from medical_image_classification_amd.gemm_ops import gemm
    M, N, K = shape
    gen = torch.Generator(device=dev()).manual_seed(M * 3 + N)
    dy = torch.randn(M, N, device=dev(), generator=gen)
    x = torch.randn(M, K, device=dev(), generator=gen)
    if M % 8 != 0:                                            # transposed reads take 8 consecutive rows: pad the token axis
        pad = 8 - M % 8
        dy = torch.cat([dy, torch.zeros(pad, N, device=dev())]); x = torch.cat([x, torch.zeros(pad, K, device=dev())])
    if not dt[0] and N % 8 == 0:
        dy = dy.to(torch.bfloat16)
    if not dt[1] and K % 8 == 0:
        x = x.to(torch.bfloat16)
    dw = gemm(dy, x, a_trans=True, b_trans=True, k_splits=splits)     # dW = dy^T x
    ref = _round(dy).t() @ _round(x)
    np.testing.assert_allclose(dw.double().cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=3e-5 * float(ref.abs().max()) + 1e-6)
    dw2 = gemm(dy, x, a_trans=True, b_trans=True, out=dw, accumulate=True)     # accumulates into what is there
    np.testing.assert_allclose(dw2.double().cpu().numpy(), 2 * ref.cpu().numpy(), rtol=0, atol=6e-5 * float(ref.abs().max()) + 1e-6)


def test_strided_rows_and_errors():
    from medical_image_classification_amd.gemm_ops import gemm
    gen = torch.Generator(device=dev()).manual_seed(1)
    big = torch.randn(300, 256, device=dev(), generator=gen)
    a = big[:, 64:160]                                        # row stride 256, 16-byte aligned column offset
    w = torch.randn(72, 96, device=dev(), generator=gen)
    y = gemm(a, w)
    ref = _round(a) @ _round(w).t()
    np.testing.assert_allclose(y.double().cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-5 * float(ref.abs().max()))
    with pytest.raises(RuntimeError):
        gemm(a, torch.randn(72, 95, device=dev()))           # inner dimensions differ
    with pytest.raises(RuntimeError):
        gemm(a.cpu(), w.cpu())                                # no CPU fallback


@pytest.mark.parametrize("cfg", [(2, 56, 56, 48, 192, True), (2, 14, 14, 192, 176, False), (1, 7, 7, 768, 384, True)])
def test_linear_mfma_autograd_vs_f_linear(cfg):
    from medical_image_classification_amd.gemm_ops import linear_mfma
    B, H, W, K, N, bf = cfg
    gen = torch.Generator(device=dev()).manual_seed(K)
    x = torch.randn(B, H, W, K, device=dev(), generator=gen)
    if bf:
        x = x.to(torch.bfloat16)
    x = x.requires_grad_()
    w = (torch.randn(N, K, device=dev(), generator=gen) * K ** -0.5).requires_grad_()
    g = torch.randn(B, H, W, N, device=dev(), generator=gen)
    y = linear_mfma(x, w, out_fp32=True)
    y.backward(g)
    x64 = _round(x.detach()).requires_grad_(); w64 = _round(w.detach()).requires_grad_()
    y64 = torch.nn.functional.linear(x64, w64)
    y64.backward(g.double())
    sc = lambda t: float(t.abs().max())
    np.testing.assert_allclose(y.detach().double().cpu().numpy(), y64.detach().cpu().numpy(), rtol=0, atol=2e-5 * sc(y64.detach()))
    # the gradients see dy rounded to bf16 (what the reference's autocast GEMMs do as well): 2^-9 relative per term
    np.testing.assert_allclose(x.grad.double().cpu().numpy(), x64.grad.cpu().numpy(), rtol=0, atol=8e-3 * sc(x64.grad))
    np.testing.assert_allclose(w.grad.double().cpu().numpy(), w64.grad.cpu().numpy(), rtol=0, atol=8e-3 * sc(w64.grad))


@pytest.mark.parametrize("shape", [(3136 * 2, 48, 96), (784, 96, 192), (392, 192, 48), (200, 36, 140)])
def test_weight_grad_orientation(shape):
    """weight_grad puts the taller of (N, K) on the row side and accumulates into the transposed output when it swaps."""
    from medical_image_classification_amd.gemm_ops import weight_grad
    M, N, K = shape
    gen = torch.Generator(device=dev()).manual_seed(N)
    dy = torch.randn(M, N, device=dev(), generator=gen); x = torch.randn(M, K, device=dev(), generator=gen)
    dw = weight_grad(dy, x)
    ref = _round(dy).t() @ _round(x)
    np.testing.assert_allclose(dw.double().cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=3e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("cfg", [(200704 // 8, 48, 48), (3136, 384, 384), (777, 136, 24), (130, 96, 96), (64 * 49, 192, 64)])
@pytest.mark.parametrize("dy_f32", [False, True])
def test_weight_gradient_with_bias_gradient_in_one_launch(cfg, dy_f32):
    """ms_gemm_bf16_wgrad_bias: dW (N, K) += dy^T x and dbias (N) += column sums of dy (one more MFMA per dy fragment against an
    all-ones fragment) vs float64 products of the bf16-rounded operands; ragged M / N / K, several k-slices."""
    from medical_image_classification_amd.gemm_ops import weight_grad
    M, N, K = cfg
    torch.manual_seed(M + N)
    dy = torch.randn(M, N, device=dev())
    x = torch.randn(M, K, device=dev()).to(torch.bfloat16)
    if not dy_f32:
        dy = dy.to(torch.bfloat16)
    dyr, xr = dy.to(torch.bfloat16).double(), x.double()
    db = torch.zeros(N, device=dev())
    dw = weight_grad(dy, x, dbias=db)
    want_w, want_b = dyr.t() @ xr, dyr.sum(0)
    np.testing.assert_allclose(dw.cpu().numpy(), want_w.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(want_w.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), want_b.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(want_b.abs().max()))
    # accumulating: a second call doubles both
    weight_grad(dy, x, out=dw, dbias=db)
    np.testing.assert_allclose(db.cpu().numpy(), 2 * want_b.cpu().numpy(), rtol=1e-4, atol=2e-4 * float(want_b.abs().max()))
    np.testing.assert_allclose(dw.cpu().numpy(), 2 * want_w.cpu().numpy(), rtol=1e-4, atol=2e-4 * float(want_w.abs().max()))


# ---- ms_gemm_f32: the same products in the reference's own precision (exact fp32 MFMA) -------------------------------------------
F32_SHAPES = [(3136 * 2, 192, 48), (784 * 2, 152, 192), (196 * 2, 768, 192), (49 * 2, 384, 768), (200, 140, 96), (129, 68, 40),
              (1, 4, 8), (1000, 48, 96), (256, 224, 768), (77, 36, 100), (3136 * 8, 140, 96)]


@pytest.mark.parametrize("shape", F32_SHAPES, ids=[f"{s[0]}x{s[1]}x{s[2]}" for s in F32_SHAPES])
def test_f32_gemm_all_products_vs_float64(shape):
    """ms_gemm_f32 (v_mfma_f32_16x16x4_f32: fp32 products, fp32 accumulation) against float64 products of the same fp32 operands:
    forward y = x W^T, input gradient dx = dy W, split-K weight gradient dW = dy^T x in both orientations, the bias + ReLU epilogue,
    strided rows.  Tolerance: fp32 accumulation over K (resp. M) terms, 4e-6 x sqrt(terms / 64) of the result's magnitude."""
    from medical_image_classification_amd.gemm_ops import gemm_f32, weight_grad_f32
    M, N, K = shape
    gen = torch.Generator(device=dev()).manual_seed(M + 7 * N + K)
    Kp, Np = (K + 3) // 4 * 4 + 4, (N + 3) // 4 * 4         # row strides: multiples of 4 floats, wider than the rows (strided views)
    xa = torch.randn(M, Kp, device=dev(), generator=gen)
    x = xa[:, :K]
    w = (torch.randn(N, Kp, device=dev(), generator=gen) * K ** -0.5)[:, :K]
    bias = torch.randn(N, device=dev(), generator=gen)
    tol = lambda ref, terms: 4e-6 * max(1.0, (terms / 64) ** 0.5) * float(ref.abs().max())
    y = gemm_f32(x, w)
    ref = x.double() @ w.double().t()
    np.testing.assert_allclose(y.double().cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=tol(ref, K))
    yb = gemm_f32(x, w, bias=bias, relu=True)
    refb = torch.relu(ref + bias.double())
    np.testing.assert_allclose(yb.double().cpu().numpy(), refb.cpu().numpy(), rtol=0, atol=tol(ref, K))
    dy = torch.randn(M, Np + 4, device=dev(), generator=gen)[:, :N]
    if N % 4 == 0:
        dx = gemm_f32(dy, w.contiguous() if K % 4 else w, b_trans=True)
        refx = dy.double() @ w.double()
        np.testing.assert_allclose(dx.double().cpu().numpy(), refx.cpu().numpy(), rtol=0, atol=tol(refx, N))
    if M % 4 == 0 and N % 4 == 0 and K % 4 == 0:
        refw = dy.double().t() @ x.double()
        for flip in (False, True):                            # both orientations of the split-K weight gradient
            if flip:
                dwt = weight_grad_f32(x, dy)                  # (K, N) = x^T dy
                np.testing.assert_allclose(dwt.double().cpu().numpy(), refw.t().cpu().numpy(), rtol=0, atol=tol(refw, M))
            else:
                dw = weight_grad_f32(dy, x)
                np.testing.assert_allclose(dw.double().cpu().numpy(), refw.cpu().numpy(), rtol=0, atol=tol(refw, M))


def test_f32_gemm_rejects_what_it_does_not_build():
    from medical_image_classification_amd import _lib
    lib = _lib.lib()
    a = torch.zeros(8, 8, device=dev()); st = _lib.current_stream_ptr(dev())
    p = a.data_ptr()
    assert lib.ms_gemm_f32(None, 0, 8, p, 0, 8, p, 0, 8, 8, 8, 8, 1, None, 0, st) == -1          # NULL
    assert lib.ms_gemm_f32(p, 0, 8, p, 0, 8, p, 1, 8, 8, 8, 8, 1, None, 0, st) == -2             # c_mode 1 (bf16 store) does not exist here
    assert lib.ms_gemm_f32(p, 0, 8, p, 0, 8, p, 0, 8, 8, 8, 8, 2, None, 0, st) == -2             # split-K needs an accumulating mode
    assert lib.ms_gemm_f32(p, 1, 8, p, 0, 8, p, 2, 8, 8, 8, 8, 1, None, 0, st) == -6             # (transposed, plain) is not built
    assert lib.ms_gemm_f32(p, 0, 6, p, 0, 8, p, 0, 8, 8, 8, 8, 1, None, 0, st) == -4             # lda not a multiple of 4
    assert lib.ms_gemm_f32(p + 4, 0, 8, p, 0, 8, p, 0, 8, 8, 8, 4, 1, None, 0, st) == -4         # unaligned base


def test_f32_linear_autograd_matches_f_linear_and_fp32_model_uses_it(monkeypatch):
    """linear_splitk without autocast = ms_gemm_f32 forward / dx / dW (vs F.linear autograd in float64); and an fp32 SS2D block runs
    NO library GEMM at all (the x_proj of the one-node inner path included): aten mm / addmm / bmm never dispatched."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from medical_image_classification_amd import medmamba as mm
    from medical_image_classification_amd.ss2d_ops import linear_splitk
    torch.manual_seed(4)
    x = torch.randn(3, 14, 14, 96, device=dev(), requires_grad=True)
    w = (torch.randn(192, 96, device=dev()) * 0.1).requires_grad_()
    g = torch.randn(3, 14, 14, 192, device=dev())
    y = linear_splitk(x, w)
    y.backward(g)
    xr, wr = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    yr = torch.nn.functional.linear(xr, wr)
    yr.backward(g.double())
    for got, want in ((y, yr), (x.grad, xr.grad), (w.grad, wr.grad)):
        np.testing.assert_allclose(got.detach().double().cpu().numpy(), want.detach().cpu().numpy(), rtol=0, atol=2e-5 * float(want.abs().max()))

    class Audit(TorchDispatchMode):
        def __init__(self):
            super().__init__(); self.seen = []
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            n = func.__name__.split(".")[0]
            if n in ("mm", "addmm", "bmm", "baddbmm", "matmul", "linear"):
                self.seen.append(n)
            return func(*args, **(kwargs or {}))

    blk = mm.SS2D(d_model=48).to(dev())
    xi = torch.randn(2, 14, 14, 48, device=dev(), requires_grad=True)
    with Audit() as a:
        out = blk(xi)
        out.sum().backward()
    assert not a.seen, a.seen


# ---- both backward products of a projection from one pass (csrc/linear_bwd.hip, ABI v9) ----------------------------------------
@pytest.mark.parametrize("M,N,K,dy_dt,x_dt,dx_dt", [
    (20000, 140, 96, torch.float32, torch.float32, torch.float32),        # x_proj, MedMamba-T stage 0 (ragged N: 4 (R + 2 * 16) = 140)
    (17000, 192, 48, torch.bfloat16, torch.bfloat16, torch.bfloat16),     # in_proj
    (16500, 48, 96, torch.float32, torch.bfloat16, torch.bfloat16),       # out_proj (N padded to 64 inside the kernel)
    (16400, 144, 128, torch.float32, torch.float32, torch.float32),       # MedMamba-B stage 0
    (16384, 256, 64, torch.bfloat16, torch.bfloat16, torch.float32),
    (16390, 64, 128, torch.bfloat16, torch.float32, torch.bfloat16),
    (70, 140, 96, torch.float32, torch.float32, torch.float32),           # one partial slab (rows threshold lowered below)
])
def test_linear_backward_both_products_from_one_pass(M, N, K, dy_dt, x_dt, dx_dt, monkeypatch):
    """ms_linear_bwd_bf16 vs float64 products of the SAME bf16-rounded operands (what the MFMA multiplies): dx to 1e-5 (+ the bf16 store's
    2^-8 when it is written in bf16), dW to 2e-5 of its max-norm (fp32 accumulation over M tokens, atomics in any order); and vs the
    two-launch form (ms_gemm_bf16 twice), which must agree to the same bounds."""
    from medical_image_classification_amd import gemm_ops
    monkeypatch.setattr(gemm_ops, "_FUSED_BWD_MIN_ROWS", 1)
    monkeypatch.setattr(gemm_ops, "_FUSED_BWD_MIN_ROW_BYTES", 0)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(M + N)
    dy = (torch.randn(M, N, device=dev, generator=g) * 0.5).to(dy_dt)
    x = (torch.randn(M, K, device=dev, generator=g) + 0.3).to(x_dt)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5)
    wb = w.to(torch.bfloat16)
    for wt in (w, wb):
        assert gemm_ops.linear_bwd_fused_ok(dy, x, wt)
        dx, dw = gemm_ops.linear_bwd_fused(dy, x, wt, dx_dt)
        torch.cuda.synchronize()
        r = lambda t: t.to(torch.bfloat16).double()
        dx_ref, dw_ref = r(dy) @ r(wt), r(dy).t() @ r(x)
        tol_dx = 1e-5 + (2 ** -8 if dx_dt == torch.bfloat16 else 0.0)
        assert float((dx.double() - dx_ref).abs().max()) <= tol_dx * float(dx_ref.abs().max()), "dx"
        assert float((dw.double() - dw_ref).abs().max()) <= 2e-5 * float(dw_ref.abs().max()), "dW"
        assert dw.dtype == torch.float32 and dw.shape == (N, K) and dx.dtype == dx_dt
    dx2 = gemm_ops.gemm(dy, wb, b_trans=True, out_dtype=dx_dt)
    dw2 = gemm_ops.weight_grad(dy, x)
    assert float((dx.double() - dx2.double()).abs().max()) <= 2 * tol_dx * float(dx_ref.abs().max())
    assert float((dw.double() - dw2.double()).abs().max()) <= 4e-5 * float(dw_ref.abs().max())


def test_linear_backward_fused_gate_and_errors():
    from medical_image_classification_amd import _lib, gemm_ops
    lib = _lib.lib()
    assert lib.ms_linear_bwd_ok(140, 96) == 1 and lib.ms_linear_bwd_ok(192, 48) == 1 and lib.ms_linear_bwd_ok(48, 96) == 1
    assert lib.ms_linear_bwd_ok(384, 96) == 0 and lib.ms_linear_bwd_ok(140, 100) == 0 and lib.ms_linear_bwd_ok(0, 96) == 0
    dev = torch.device("cuda:0")
    dy, x, w = torch.zeros(20000, 140, device=dev), torch.zeros(20000, 96, device=dev), torch.zeros(140, 96, device=dev)
    assert gemm_ops.linear_bwd_fused_ok(dy, x, w)
    assert not gemm_ops.linear_bwd_fused_ok(dy[:1000], x[:1000], w)                  # few rows: the two-launch form
    assert not gemm_ops.linear_bwd_fused_ok(torch.zeros(20000, 48, device=dev), x.to(torch.bfloat16), torch.zeros(48, 96, device=dev))   # short rows (384 B): likewise
    assert not gemm_ops.linear_bwd_fused_ok(dy, x, torch.zeros(384, 96, device=dev))
    assert not gemm_ops.linear_bwd_fused_ok(dy[:, 1:], x, w[1:])                     # misaligned rows
    st = _lib.current_stream_ptr(dev)
    assert lib.ms_linear_bwd_bf16(None, 1, 140, x.data_ptr(), 1, 96, w.data_ptr(), 1, x.data_ptr(), 0, 96, w.data_ptr(), 20000, 140, 96, st) == -1
    assert lib.ms_linear_bwd_bf16(dy.data_ptr(), 1, 140, x.data_ptr(), 1, 96, w.data_ptr(), 1, x.data_ptr(), 0, 96, w.data_ptr(), 20000, 384, 96, st) == -6
    assert lib.ms_linear_bwd_bf16(dy.data_ptr(), 1, 141, x.data_ptr(), 1, 96, w.data_ptr(), 1, x.data_ptr(), 0, 96, w.data_ptr(), 20000, 140, 96, st) == -4
