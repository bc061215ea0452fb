"""FusionMamba's S6 fusion blocks on the HIP scan kernels (CrossMamba/FusionMamba/models/cross.py:417-1384):
the second consumer of the selective-scan operator in the reference.  Same class names, constructor arguments, forward
signatures and state_dict keys as cross.py:

    SS2D                  cross.py:417-742    in_proj -> depthwise conv + act -> stride-2 four-way selective scan -> * z -> out_proj
    SS2D_cross_new        cross.py:890-1230   two modalities; the scan sees x1*x2 + x1 + x2, the result gates both z's
    VSSBlock_new          cross.py:1297-1384  norm -> SS2D -> ECA / LDC conv branch / BiAttn mixing -> residual (+ MLP)
    VSSBlock_Cross_new    cross.py:1262-1295  texture / differential enhancement of both inputs -> SS2D_cross_new -> ECA -> residual
    BiAttn, Mlp, LDC, Enhancement_texture_LDC, Differential_enhance, Cross_layer, eca_layer  (:744-888, :1233-1260)

The scan is `efficient_scan.cross_selective_scan*`: on CUDA the four stride-2 sub-lattices are an ADDRESSING MODE of the scan
kernels (MS_SCAN_LATTICE) -- no gathered sequences, no merge pass.  Built forward types: "v1" / "v2" (the reference's default and
the only ones its models construct) with their "nozact" / "softmax" / "sigmoid" suffixes; "v0", "v0_seq", "share_ssm", "share_a"
(cross.py:598-707, debugging variants of the upstream VMamba code) raise NotImplementedError.
Unlike cross.py nothing calls `.cuda()` inside a constructor (cross.py:800,826): the modules follow `.to(device)`.

Reference quirks kept, because weights trained with the reference depend on them:
  * SS2D_cross_new.forward computes z2 = act2(z1) -- from the ALREADY ACTIVATED z1, not from its own z2 (cross.py:1209-1210);
  * Differential_enhance owns a `lastconv` it never applies (cross.py:849).
"""
import math
from functools import partial
from typing import Any, Callable

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

from .efficient_scan import cross_selective_scan, cross_selective_scan_cross
from .medmamba import DropPath

_STUBS = ("share_ssm", "share_a")      # cores whose reference bodies are `...` (cross.py:697-707): nothing to build


class _S6Mixer(nn.Module):
    """Parameters shared by SS2D and SS2D_cross_new (cross.py:417-541 and its copy at :890-1016): everything but the input
    projection(s).  Attribute names are the reference's state_dict keys."""

    def _build(self, d_model, d_state, ssm_ratio, ssm_rank_ratio, dt_rank, d_conv, conv_bias, dropout, bias, dt_min, dt_max, dt_init,
               dt_scale, dt_init_floor, simple_init, forward_type, step_size):
        d_expand = int(ssm_ratio * d_model)
        d_inner = int(min(ssm_rank_ratio, ssm_ratio) * d_model) if ssm_rank_ratio > 0 else d_expand
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.d_state = math.ceil(d_model / 6) if d_state == "auto" else d_state
        self.d_conv, self.step_size = d_conv, step_size
        # forward_type = core name + optional suffixes (cross.py:455-483)
        self.disable_z_act = forward_type.endswith("nozact")
        if self.disable_z_act:
            forward_type = forward_type[:-len("nozact")]
        if forward_type.endswith("softmax"):
            forward_type, self.out_norm = forward_type[:-len("softmax")], nn.Softmax(dim=1)
        elif forward_type.endswith("sigmoid"):
            forward_type, self.out_norm = forward_type[:-len("sigmoid")], nn.Sigmoid()
        else:
            self.out_norm = nn.LayerNorm(d_inner)
        # cores (cross.py:475-485): v1 / v2 = the stride-2 scan; v0 / v0_seq = the full-resolution four-direction scan of VMamba v0
        # (= MedMamba's forward_corev0 followed by out_norm), built on this package's SS2D scan core; share_ssm / share_a are `...` in
        # the reference (they return None): their parameter shapes (K, K2) are kept, calling them raises
        self.core_type = forward_type if forward_type in ("v0", "v0_seq", "v1", "v2") + _STUBS else "v2"
        self.K = 1 if forward_type == "share_ssm" else 4
        self.K2 = 1 if forward_type == "share_a" else self.K
        if self.d_conv > 1:
            self.conv2d = nn.Conv2d(d_expand, d_expand, groups=d_expand, bias=conv_bias, kernel_size=d_conv, padding=(d_conv - 1) // 2)
        self.ssm_low_rank = d_inner < d_expand
        if self.ssm_low_rank:
            self.in_rank = nn.Conv2d(d_expand, d_inner, kernel_size=1, bias=False)
            self.out_rank = nn.Linear(d_inner, d_expand, bias=False)
        width = self.dt_rank + 2 * self.d_state
        self.x_proj_weight = nn.Parameter(torch.stack([nn.Linear(d_inner, width, bias=False).weight for _ in range(self.K)], dim=0))
        dts = [self.dt_init(self.dt_rank, d_inner, dt_scale, dt_init, dt_min, dt_max, dt_init_floor) for _ in range(self.K)]
        self.dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in dts], dim=0))      # (K, inner, rank)
        self.dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in dts], dim=0))          # (K, inner)
        self.A_logs = self.A_log_init(self.d_state, d_inner, copies=self.K2, merge=True)      # (K * inner, N)
        self.Ds = self.D_init(d_inner, copies=self.K2, merge=True)                            # (K * inner)
        self.out_proj = nn.Linear(d_expand, d_model, bias=bias)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        if simple_init:
            self.Ds = nn.Parameter(torch.ones(self.K2 * d_inner))
            self.A_logs = nn.Parameter(torch.randn(self.K2 * d_inner, self.d_state))
            self.dt_projs_weight = nn.Parameter(torch.randn(self.K, d_inner, self.dt_rank))
            self.dt_projs_bias = nn.Parameter(torch.randn(self.K, d_inner))
        return d_expand

    @staticmethod
    def dt_init(dt_rank, d_inner, dt_scale=1.0, dt_init="random", dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4, **factory_kwargs):
        """A Linear(dt_rank, d_inner) whose bias is the softplus inverse of a log-uniform dt in [dt_min, dt_max] (cross.py:543-566)."""
        proj = nn.Linear(dt_rank, d_inner, bias=True, **factory_kwargs)
        std = dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(proj.weight, std)
        elif dt_init == "random":
            nn.init.uniform_(proj.weight, -std, std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(d_inner, **factory_kwargs) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
        with torch.no_grad():
            proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
        return proj

    @staticmethod
    def A_log_init(d_state, d_inner, copies=-1, device=None, merge=True):
        """log of the S4D-real spectrum 1..d_state per channel (cross.py:568-583)."""
        a = torch.log(torch.arange(1, d_state + 1, dtype=torch.float32, device=device)).repeat(d_inner, 1)
        if copies > 0:
            a = a.unsqueeze(0).repeat(copies, 1, 1)
            if merge:
                a = a.flatten(0, 1)
        a = nn.Parameter(a.contiguous())
        a._no_weight_decay = True
        return a

    @staticmethod
    def D_init(d_inner, copies=-1, device=None, merge=True):
        d = torch.ones(d_inner, device=device)
        if copies > 0:
            d = d.unsqueeze(0).repeat(copies, 1)
            if merge:
                d = d.flatten(0, 1)
        d = nn.Parameter(d)
        d._no_weight_decay = True
        return d

    def forward_corev0(self, x, to_dtype=False, channel_first=False, **_ignored):
        """cross.py:598-646: cross-scan (4 directions at full resolution) -> x_proj / dt_proj -> selective scan -> cross-merge ->
        out_norm; here on the SS2D addressing mode of the scan kernels (nothing is permuted or copied).  The reference's own forward()
        cannot reach this core -- it passes `step_size=`, which the v0 signatures do not take (TypeError at cross.py:732) -- so extra
        keywords are accepted and ignored; called directly, as the reference allows, the behaviour is the same."""
        from .ss2d_fused import ss2d_core
        if channel_first:
            x = x.permute(0, 2, 3, 1)
        B, H, W, D = x.shape
        y = ss2d_core(x.contiguous().float(), self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias, self.A_logs, self.Ds,
                      self.d_state, self.dt_rank)                                  # (B, H, W, D) merged, fp32
        y = self.out_norm(y.view(B, H * W, D)).view(B, H, W, -1)                   # LayerNorm over channels (Softmax: over dim 1 = L)
        return y.to(x.dtype) if to_dtype else y

    # one selective_scan_fn call per direction in the reference (cross.py:648-695): the same function of the same operands.  (The
    # reference's own v0_seq cannot run: its wrapper passes delta_bias in selective_scan_fn's `z` slot, cross.py:650.)
    forward_corev0_seq = forward_corev0

    def _forward_core_stub(self, *a, **k):
        raise NotImplementedError(f"forward_type {self.core_type!r}: the reference's body is `...` (cross.py:697-707)")

    def _pick_core(self, v2):
        return {"v0": self.forward_corev0, "v0_seq": self.forward_corev0_seq, "share_ssm": self._forward_core_stub,
                "share_a": self._forward_core_stub}.get(self.core_type, v2)

    def _conv_act(self, x, act):
        """(b,h,w,d) -> act(conv2d(x)) as (b,d,h,w) (cross.py:731-732)."""
        return act(self.conv2d(x.permute(0, 3, 1, 2).contiguous()))

    def _scan_args(self):
        return (self.x_proj_weight, None, self.dt_projs_weight, self.dt_projs_bias, self.A_logs, self.Ds, getattr(self, "out_norm", None))


class SS2D(_S6Mixer):
    def __init__(self, d_model=96, d_state=16, ssm_ratio=2.0, ssm_rank_ratio=2.0, dt_rank="auto", act_layer=nn.SiLU,
                 d_conv=3, conv_bias=True, dropout=0.0, bias=False, dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0,
                 dt_init_floor=1e-4, simple_init=False, forward_type="v2", step_size=2, **kwargs):
        super().__init__()
        d_expand = int(ssm_ratio * d_model)
        self.in_proj = nn.Linear(d_model, d_expand * 2, bias=bias)
        self.act = act_layer()
        self._build(d_model, d_state, ssm_ratio, ssm_rank_ratio, dt_rank, d_conv, conv_bias, dropout, bias, dt_min, dt_max, dt_init,
                    dt_scale, dt_init_floor, simple_init, forward_type, step_size)
        self.forward_core = self._pick_core(self.forward_corev2)

    def forward_corev2(self, x, nrows=-1, channel_first=False, step_size=2):
        if not channel_first:
            x = x.permute(0, 3, 1, 2).contiguous()
        if self.ssm_low_rank:
            x = self.in_rank(x)
        x = cross_selective_scan(x, *self._scan_args(), nrows=1, delta_softplus=True, step_size=step_size)
        return self.out_rank(x) if self.ssm_low_rank else x

    def forward(self, x, **kwargs):
        xz = self.in_proj(x)
        if self.d_conv > 1:
            x, z = xz.chunk(2, dim=-1)
            if not self.disable_z_act:
                z = self.act(z)
            x = self._conv_act(x, self.act)
        elif self.disable_z_act:
            x, z = xz.chunk(2, dim=-1)
            x = self.act(x)
        else:
            x, z = self.act(xz).chunk(2, dim=-1)
        y = self.forward_core(x, channel_first=(self.d_conv > 1), step_size=self.step_size)
        return self.dropout(self.out_proj(y * z))


class SS2D_cross_new(_S6Mixer):
    def __init__(self, d_model=96, d_state=16, ssm_ratio=2.0, ssm_rank_ratio=2.0, dt_rank="auto", act_layer=nn.SiLU,
                 d_conv=3, conv_bias=True, dropout=0.0, bias=False, dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0,
                 dt_init_floor=1e-4, simple_init=False, forward_type="v2", step_size=2, **kwargs):
        super().__init__()
        d_expand = int(ssm_ratio * d_model)
        self.in_proj1 = nn.Linear(d_model, d_expand * 2, bias=bias)
        self.in_proj2 = nn.Linear(d_model, d_expand * 2, bias=bias)
        self.act1, self.act2 = act_layer(), act_layer()
        self._build(d_model, d_state, ssm_ratio, ssm_rank_ratio, dt_rank, d_conv, conv_bias, dropout, bias, dt_min, dt_max, dt_init,
                    dt_scale, dt_init_floor, simple_init, forward_type, step_size)
        self.forward_core = self.forward_corev2

    def forward_corev2(self, x1, x2, nrows=-1, channel_first=False, step_size=2):
        if not channel_first:
            x1, x2 = x1.permute(0, 3, 1, 2).contiguous(), x2.permute(0, 3, 1, 2).contiguous()
        if self.ssm_low_rank:
            x1, x2 = self.in_rank(x1), self.in_rank(x2)
        x = cross_selective_scan_cross(x1, x2, *self._scan_args(), nrows=1, delta_softplus=True, step_size=step_size)
        return self.out_rank(x) if self.ssm_low_rank else x

    def forward(self, x1, x2, **kwargs):
        xz1, xz2 = self.in_proj1(x1), self.in_proj2(x2)
        if self.d_conv > 1:
            x1, z1 = xz1.chunk(2, dim=-1)
            x2, z2 = xz2.chunk(2, dim=-1)
            if not self.disable_z_act:
                z1 = self.act1(z1)
                z2 = self.act2(z1)            # sic: the reference activates z1 twice and drops its own z2 (cross.py:1209-1210)
            x1, x2 = self._conv_act(x1, self.act1), self._conv_act(x2, self.act2)      # ONE depthwise conv serves both modalities
        elif self.disable_z_act:
            x1, z1 = xz1.chunk(2, dim=-1)
            x2, z2 = xz2.chunk(2, dim=-1)
            x1, x2 = self.act1(x1), self.act2(x2)
        else:
            x1, z1 = self.act1(xz1).chunk(2, dim=-1)
            x2, z2 = self.act2(xz2).chunk(2, dim=-1)
        y = self.forward_core(x1, x2, channel_first=(self.d_conv > 1), step_size=self.step_size)
        return self.dropout(self.out_proj(y * z1 + y * z2))


class BiAttn(nn.Module):
    """Channel attention from the LayerNormed global mean (cross.py:744-767); x is (b,h,w,c)."""

    def __init__(self, in_channels, act_ratio=0.125, act_fn=nn.GELU, gate_fn=nn.Sigmoid):
        super().__init__()
        reduce_channels = int(in_channels * act_ratio)
        self.norm = nn.LayerNorm(in_channels)
        self.global_reduce = nn.Linear(in_channels, reduce_channels)
        self.act_fn = act_fn()
        self.channel_select = nn.Linear(reduce_channels, in_channels)
        self.gate_fn = gate_fn()

    def forward(self, x):
        pooled = self.norm(x).mean([1, 2], keepdim=True)
        return x * self.gate_fn(self.channel_select(self.act_fn(self.global_reduce(pooled))))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, channels_first=False):
        super().__init__()
        out_features, hidden_features = out_features or in_features, hidden_features or in_features
        linear = partial(nn.Conv2d, kernel_size=1, padding=0) if channels_first else nn.Linear
        self.fc1 = linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class LDC(nn.Module):
    """Learnable difference convolution (cross.py:790-813): a 3x3 convolution whose centre tap is reduced by
    theta * mask[o,i] * (sum of the 3x3 kernel) -- an input-difference term folded into the weights."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1, bias=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation,
                              groups=groups, bias=bias)
        # cross.py keeps this as a plain `.cuda()` tensor attribute (not in the state_dict): a non-persistent buffer follows .to()
        self.register_buffer("center_mask", torch.tensor([[0, 0, 0], [0, 1, 0], [0, 0, 0]]), persistent=False)
        self.base_mask = nn.Parameter(torch.ones(self.conv.weight.size()), requires_grad=False)
        self.learnable_mask = nn.Parameter(torch.ones([self.conv.weight.size(0), self.conv.weight.size(1)]), requires_grad=True)
        self.learnable_theta = nn.Parameter(torch.ones(1) * 0.5, requires_grad=True)

    def forward(self, x):
        w = self.conv.weight
        mask = self.base_mask - self.learnable_theta * self.learnable_mask[:, :, None, None] * self.center_mask * w.sum(2).sum(2)[:, :, None, None]
        return F.conv2d(input=x, weight=w * mask, bias=self.conv.bias, stride=self.conv.stride, padding=self.conv.padding,
                        groups=self.conv.groups)


class Enhancement_texture_LDC(LDC):
    """cross.py:816-840: the same operator under a second name."""


class Differential_enhance(nn.Module):
    def __init__(self, nf=48):
        super().__init__()
        self.global_avgpool = nn.AdaptiveAvgPool2d(1)
        self.act = nn.Sigmoid()
        self.lastconv = nn.Conv2d(nf, nf // 2, 1, 1)          # constructed but never applied by the reference either

    def forward(self, fuse, x1, x2):
        w12 = self.act(self.global_avgpool(x1 - x2))
        w21 = self.act(self.global_avgpool(x2 - x1))
        return w12 * fuse + x1, w21 * fuse + x2


class Cross_layer(nn.Module):
    def __init__(self, hidden_dim: int = 0):
        super().__init__()
        self.d_model = hidden_dim
        self.texture_enhance1 = Enhancement_texture_LDC(hidden_dim, hidden_dim)
        self.texture_enhance2 = Enhancement_texture_LDC(hidden_dim, hidden_dim)
        self.Diff_enhance = Differential_enhance(hidden_dim)

    def forward(self, Fuse, x1, x2):
        d1, d2 = self.Diff_enhance(Fuse, x1, x2)
        return self.texture_enhance1(x1) + d1, self.texture_enhance2(x2) + d2


class eca_layer(nn.Module):
    """Efficient channel attention (cross.py:1233-1260): a k-tap 1-D convolution over the channel axis of the global mean."""

    def __init__(self, channel, k_size=3):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv = nn.Conv1d(1, 1, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        y = self.conv(self.avg_pool(x).squeeze(-1).transpose(-1, -2)).transpose(-1, -2).unsqueeze(-1)
        return x * self.sigmoid(y).expand_as(x)


class VSSBlock_Cross_new(nn.Module):
    def __init__(self, hidden_dim: int = 0, drop_path: float = 0, norm_layer: Callable[..., nn.Module] = partial(nn.LayerNorm, eps=1e-6),
                 attn_drop_rate: float = 0, d_state: int = 16, **kwargs):
        super().__init__()
        self.ln_1 = norm_layer(hidden_dim)
        self.ln_2 = norm_layer(hidden_dim)
        self.Cross_layer = Cross_layer(hidden_dim)
        self.self_attention_cross = SS2D_cross_new(d_model=hidden_dim, dropout=attn_drop_rate, d_state=d_state, **kwargs)
        self.self_attention_cross_spatial = eca_layer(channel=hidden_dim)
        self.drop_path = DropPath(drop_path)                  # constructed, not applied (cross.py:1280-1295)

    def forward(self, input1, input2):
        x1, x2 = input1.permute(0, 3, 1, 2), input2.permute(0, 3, 1, 2)
        f1, f2 = self.Cross_layer(x1 + x2, x1, x2)
        cross = self.self_attention_cross(self.ln_1(f1.permute(0, 2, 3, 1)), self.ln_2(f2.permute(0, 2, 3, 1)))      # (b,h,w,c)
        spatial = self.self_attention_cross_spatial(cross.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        return input2 + input1 + cross + spatial


class VSSBlock_new(nn.Module):
    def __init__(self, hidden_dim: int = 0, drop_path: float = 0, norm_layer: Callable[..., nn.Module] = partial(nn.LayerNorm, eps=1e-6),
                 ssm_d_state: int = 16, ssm_ratio=2.0, ssm_rank_ratio=2.0, ssm_dt_rank: Any = "auto", ssm_act_layer=nn.SiLU,
                 ssm_conv: int = 3, ssm_conv_bias=True, ssm_drop_rate: float = 0, ssm_simple_init=False, forward_type="v2",
                 mlp_ratio=4.0, mlp_act_layer=nn.GELU, mlp_drop_rate: float = 0.0, use_checkpoint: bool = False, step_size=2, **kwargs):
        super().__init__()
        self.use_checkpoint = use_checkpoint
        self.norm = norm_layer(hidden_dim)
        self.op = SS2D(d_model=hidden_dim, d_state=ssm_d_state, ssm_ratio=ssm_ratio, ssm_rank_ratio=ssm_rank_ratio, dt_rank=ssm_dt_rank,
                       act_layer=ssm_act_layer, d_conv=ssm_conv, conv_bias=ssm_conv_bias, dropout=ssm_drop_rate,
                       simple_init=ssm_simple_init, forward_type=forward_type, step_size=step_size)
        self.conv_branch = LDC(hidden_dim, hidden_dim)
        self.self_attention_cross_channel = eca_layer(channel=hidden_dim)
        self.se = BiAttn(hidden_dim)
        self.drop_path = DropPath(drop_path)
        self.mlp_branch = mlp_ratio > 0
        if self.mlp_branch:
            self.norm2 = norm_layer(hidden_dim)
            self.mlp = Mlp(in_features=hidden_dim, hidden_features=int(hidden_dim * mlp_ratio), act_layer=mlp_act_layer,
                           drop=mlp_drop_rate, channels_first=False)

    def _forward(self, input):
        x_ssm = self.op(self.norm(input))
        x = x_ssm + self.self_attention_cross_channel(x_ssm.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        x_conv = self.conv_branch(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        x = input + self.drop_path(self.se(x_ssm) + self.se(x_conv))
        if self.mlp_branch:
            x = x + self.drop_path(self.mlp(self.norm2(x)))
        return x

    def forward(self, input):
        return checkpoint.checkpoint(self._forward, input) if self.use_checkpoint else self._forward(input)
