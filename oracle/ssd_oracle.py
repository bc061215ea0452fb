"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the SSD (Mamba-2) path the reference reaches through
`mamba_ssm==2.2.2` (README.md:7), a dependency that is NOT in the reference tree and not installable here.

PARITY UNPINNED: nothing in the reference holds vectors for `mamba_chunk_scan_combined` / `RMSNormGated`
(SURVEY.md 8c), and the Triton kernels cannot run in this image.  What is restated is the published recurrence at the
call-site contract of CNN_Mamba.py:506-537 (shapes, dt_softplus, dt_bias, D), written as an independent sequential loop
over l; tests/test_ssd_cpu.py checks it against the PINNED S6 oracle (oracle/scan_oracle.c) by expanding scalar-A heads
to diagonal A, which is the identity SURVEY.md 2a row a22 states.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math

import torch
import torch.nn.functional as F


def ssd_scan_ref(x, dt, A, B, C, D=None, dt_bias=None, dt_softplus=False):
    """x (b,l,h,p), dt (b,l,h), A (h), B/C (b,l,g,n), D (h)|(h,p), dt_bias (h) -> y (b,l,h,p).  float64 inside.
        h_t[h,p,n] = exp(dt_t[h] A[h]) h_{t-1}[h,p,n] + dt_t[h] B_t[g(h),n] x_t[h,p];  y_t = sum_n C_t[g(h),n] h_t + D x_t"""
    b, l, h, p = x.shape
    g, n = B.shape[2], B.shape[3]
    x64, dt64, A64 = x.double(), dt.double(), A.double()
    if dt_bias is not None:
        dt64 = dt64 + dt_bias.double()
    if dt_softplus:
        dt64 = F.softplus(dt64)
    Bh = B.double().repeat_interleave(h // g, dim=2)          # (b,l,h,n)
    Ch = C.double().repeat_interleave(h // g, dim=2)
    state = torch.zeros(b, h, p, n, dtype=torch.float64)
    ys = []
    for t in range(l):
        a = torch.exp(dt64[:, t] * A64)                       # (b,h)
        state = a[:, :, None, None] * state + (dt64[:, t, :, None] * x64[:, t])[..., None] * Bh[:, t, :, None, :]
        ys.append((state * Ch[:, t, :, None, :]).sum(-1))
    y = torch.stack(ys, dim=1)
    if D is not None:
        Dd = D.double()
        y = y + x64 * (Dd if Dd.dim() == 2 else Dd[:, None])
    return y.to(x.dtype)


def ssd_scan_by_expansion(x, dt, A, B, C, D=None, dt_bias=None, dt_softplus=False):
    """The same operator evaluated by the PINNED S6 oracle (oracle/scan_oracle.c through its autograd front-end) with every
    scalar-A head expanded to diagonal A: channel d = (head, p), A[d, :] = A[head], delta[d] = dt[head], B / C groups shared by
    h / g heads -- the identity of SURVEY.md 8(a) row a22, which tests/test_ssd_cpu.py holds to `ssd_scan_ref` (output and all
    gradients).  fp32, C / OpenMP: what makes an oracle run at 224 x 224 (L = 3136) affordable; differentiable."""
    from .scan_oracle import selective_scan_oracle
    b, l, h, p = x.shape
    g, n = B.shape[2], B.shape[3]
    dim = h * p
    u = x.float().reshape(b, l, dim).permute(0, 2, 1)
    delta = dt.float().unsqueeze(-1).expand(b, l, h, p).reshape(b, l, dim).permute(0, 2, 1)
    A6 = A.float().view(h, 1, 1).expand(h, p, n).reshape(dim, n)
    D6 = None
    if D is not None:
        D6 = D.float().reshape(dim) if D.dim() == 2 else D.float().view(h, 1).expand(h, p).reshape(dim)
    b6 = dt_bias.float().view(h, 1).expand(h, p).reshape(dim) if dt_bias is not None else None
    y = selective_scan_oracle(u.contiguous(), delta.contiguous(), A6.contiguous(), B.float().permute(0, 2, 3, 1).contiguous(),
                              C.float().permute(0, 2, 3, 1).contiguous(), D6.contiguous() if D6 is not None else None, None,
                              b6.contiguous() if b6 is not None else None, dt_softplus)
    return y.permute(0, 2, 1).reshape(b, l, h, p).to(x.dtype)


# tests at full image sizes switch the module oracles below to the expansion (install_ssd(model, by_expansion=True))
_SCAN = {"fn": ssd_scan_ref}


def rmsnorm_gated_ref(x, z, weight, eps=1e-5, norm_before_gate=False):
    x = x.double()
    if z is not None and not norm_before_gate:
        x = x * F.silu(z.double())
    y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * weight.double()
    if z is not None and norm_before_gate:
        y = y * F.silu(z.double())
    return y


def ss2d_ssd_forward_oracle(mod, u):
    """SS2D_with_SSD.forward (CNN_Mamba.py:454-564) op for op on CPU tensors, with the reference's own tensor shuffles
    (stack / transpose / flip / cat), given a module holding the reference's parameters."""
    B, H, W, C = u.shape
    L, K = H * W, 4
    GN = mod.ngroups * mod.d_state
    zxbcdt = F.linear(u, mod.in_proj.weight, mod.in_proj.bias)
    d_mlp = (zxbcdt.shape[-1] - 2 * mod.d_ssm - 2 * GN - mod.nheads) // 2
    z0, x0, z, xBCdt = torch.split(zxbcdt, [d_mlp, d_mlp, mod.d_ssm, mod.d_ssm + 2 * GN + mod.nheads], dim=-1)
    xBCdt = F.silu(F.conv2d(xBCdt.permute(0, 3, 1, 2), mod.conv2d.weight, mod.conv2d.bias,
                            padding=(mod.d_conv - 1) // 2, groups=xBCdt.shape[-1]))
    hwwh = torch.stack([xBCdt.reshape(B, -1, L), xBCdt.transpose(2, 3).reshape(B, -1, L)], dim=1)
    xBCdts = torch.cat([hwwh, hwwh.flip(-1)], dim=1)                                   # (B,4,conv_dim,L)
    xs, Bs, Cs, dts = torch.split(xBCdts, [mod.d_ssm, GN, GN, mod.nheads], dim=2)
    xs = xs.permute(0, 3, 1, 2).reshape(B, L, K * mod.nheads, mod.headdim)
    dts = dts.permute(0, 3, 1, 2).reshape(B, L, K * mod.nheads)
    Bs = Bs.reshape(B, -1, L).permute(0, 2, 1).reshape(B, L, mod.ngroups, -1)
    Cs = Cs.reshape(B, -1, L).permute(0, 2, 1).reshape(B, L, mod.ngroups, -1)
    As = -torch.exp(mod.A_logs.float())
    Ds = mod.Ds.view(-1, mod.headdim) if mod.D_has_hdim else mod.Ds
    y = _SCAN["fn"](xs.float(), dts.float(), As, Bs.float(), Cs.float(), D=Ds, dt_bias=mod.dt_bias.view(-1),
                    dt_softplus=True)
    out_y = y.reshape(B, L, K, -1).permute(0, 2, 3, 1)                                 # (B,4,d_ssm,L)
    inv_y = out_y[:, 2:4].flip(-1)
    wh_y = out_y[:, 1].reshape(B, -1, W, H).transpose(2, 3).reshape(B, -1, L)
    invwh_y = inv_y[:, 1].reshape(B, -1, W, H).transpose(2, 3).reshape(B, -1, L)
    out = out_y[:, 0] + inv_y[:, 0] + wh_y + invwh_y
    out = out.transpose(1, 2).reshape(B, H, W, -1)
    if mod.rmsnorm:
        out = rmsnorm_gated_ref(out, z, mod.norm.weight, eps=mod.norm.eps,
                                norm_before_gate=mod.norm_before_gate).to(u.dtype)
    if d_mlp > 0:
        out = torch.cat([F.silu(z0) * x0, out], dim=-1)
    return F.linear(out, mod.out_proj.weight, mod.out_proj.bias)


def _ssd_core_oracle(mod, xBCdt_nchw, z, z0, x0, d_mlp, B, H, W):
    """The part SS2D_with_SSD.forward and CrossMamba.forward.mamba_core share, with the reference's tensor shuffles."""
    L, K = H * W, 4
    GN = mod.ngroups * mod.d_state
    hwwh = torch.stack([xBCdt_nchw.reshape(B, -1, L), xBCdt_nchw.transpose(2, 3).reshape(B, -1, L)], dim=1)
    xBCdts = torch.cat([hwwh, hwwh.flip(-1)], dim=1)
    xs, Bs, Cs, dts = torch.split(xBCdts, [mod.d_ssm, GN, GN, mod.nheads], dim=2)
    xs = xs.permute(0, 3, 1, 2).reshape(B, L, K * mod.nheads, mod.headdim)
    dts = dts.permute(0, 3, 1, 2).reshape(B, L, K * mod.nheads)
    Bs = Bs.reshape(B, -1, L).permute(0, 2, 1).reshape(B, L, mod.ngroups, -1)
    Cs = Cs.reshape(B, -1, L).permute(0, 2, 1).reshape(B, L, mod.ngroups, -1)
    As = -torch.exp(mod.A_logs.float())
    Ds = mod.Ds.view(-1, mod.headdim) if mod.D_has_hdim else mod.Ds
    y = _SCAN["fn"](xs.float(), dts.float(), As, Bs.float(), Cs.float(), D=Ds, dt_bias=mod.dt_bias.view(-1),
                    dt_softplus=True)
    out_y = y.reshape(B, L, K, -1).permute(0, 2, 3, 1)
    inv_y = out_y[:, 2:4].flip(-1)
    wh_y = out_y[:, 1].reshape(B, -1, W, H).transpose(2, 3).reshape(B, -1, L)
    invwh_y = inv_y[:, 1].reshape(B, -1, W, H).transpose(2, 3).reshape(B, -1, L)
    out = out_y[:, 0] + inv_y[:, 0] + wh_y + invwh_y
    out = out.transpose(1, 2).reshape(B, H, W, -1)
    if mod.rmsnorm:
        out = rmsnorm_gated_ref(out, z, mod.norm.weight, eps=mod.norm.eps, norm_before_gate=mod.norm_before_gate).to(z.dtype)
    if d_mlp > 0:
        out = torch.cat([F.silu(z0) * x0, out], dim=-1)
    return F.linear(out, mod.out_proj.weight, mod.out_proj.bias)


def crossmamba_forward_oracle(mod, u1, u2, u2_cat_u1, u1_cat_u2):
    """CrossMamba.forward (CrossMamba_fusion_2b2.py:235-385) on CPU tensors, op for op."""
    B, H, W, _ = u1.shape

    def dw(conv, t):
        return F.silu(F.conv2d(t.permute(0, 3, 1, 2), conv.weight, conv.bias, padding=(mod.d_conv - 1) // 2, groups=t.shape[-1]))

    def one(u, up):
        zx = F.linear(u, mod.skip_in_proj.weight, mod.skip_in_proj.bias)
        d_mlp = (zx.shape[-1] - mod.d_ssm) // 2
        z0, x0, z = torch.split(zx, [d_mlp, d_mlp, mod.d_ssm], dim=-1)
        xs = dw(mod.xs_conv2d, F.linear(u, mod.xs_in_proj.weight, mod.xs_in_proj.bias))
        bcd = dw(mod.BCdts_conv2d, F.linear(up, mod.BCdts_in_proj.weight, mod.BCdts_in_proj.bias))
        return _ssd_core_oracle(mod, torch.cat([xs, bcd], dim=1), z, z0, x0, d_mlp, B, H, W)

    return one(u1, u2_cat_u1), one(u2, u1_cat_u2)


def install_ssd(model, by_expansion=False):
    """Rebind every SS2D_with_SSD and CrossMamba in `model` to the CPU restatement (CPU-side checker of the GPU modules).
    by_expansion: the scans of `_ssd_core_oracle` (CrossMamba, and the SS_Conv_SSD blocks routed through it) run on the pinned
    C oracle by expansion instead of the float64 Python loop -- process-wide until the next install_ssd call."""
    import types
    _SCAN["fn"] = ssd_scan_by_expansion if by_expansion else ssd_scan_ref
    from medical_image_classification_amd.cnn_mamba import SS2D_with_SSD
    from medical_image_classification_amd.crossmamba import CrossMamba
    for m in model.modules():
        if isinstance(m, SS2D_with_SSD):
            m.forward = types.MethodType(lambda self, u, **kw: ss2d_ssd_forward_oracle(self, u), m)
        elif isinstance(m, CrossMamba):
            m.forward = types.MethodType(lambda self, u1, u2, c21, c12, **kw: crossmamba_forward_oracle(self, u1, u2, c21, c12), m)
    return model


def vfefm_forward_oracle(model, x1, x2):
    """VFEFM.forward (CrossMamba_fusion_2b2.py:1276-1285) on CPU for a model prepared by install_ssd: the product's forward
    refuses CPU tensors, its stage methods are plain torch once the SSD modules are rebound."""
    f1, f2, skip = model.forward_down(x1.permute(0, 2, 3, 1), x2.permute(0, 2, 3, 1))
    return model.final_conv(model.forward_up(f1, f2, skip).permute(0, 3, 1, 2))
