"""Cross-modal SSD fusion block -- module surface of the reference's CrossMamba/CrossMamba_fusion_2b2.py:54-385
(`CrossMamba`; `MedSSD`, :390, is `cnn_mamba.SS2D_with_SSD`): two SSD scans over two modalities' feature maps in which the
content-aware parameters (B, C, dt) of each modality come from a projection of the OTHER pairing
(`u2_cat_u1` drives modality 1, `u1_cat_u2` drives modality 2, :263-268), sharing one set of weights.
Same constructor arguments, forward signature and state_dict keys (including `in_proj` and `conv2d`, which the reference
constructs but never uses in forward, :121-146).  Scan, conv and norm run on the same HIP kernels as SS2D_with_SSD
(cnn_mamba.py); PARITY UNPINNED against the Triton dependency exactly as there.  `PyTorchModelHubMixin` (hub
upload/download helpers) is not part of the compute path and is not mixed in.
"""
import math

import torch
import torch.nn as nn

from . import _lib
from .cnn_mamba import RMSNormGated, SS2D_with_SSD, proj, ssd_dwconv_silu, ssd_scan_merge, ssd_tail

MedSSD = SS2D_with_SSD


class CrossMamba(nn.Module):
    def __init__(self, d_model, d_state=128, d_conv=3, expand=2, headdim=64, d_ssm=None, ngroups=1,
                 A_init_range=(1, 16), D_has_hdim=False, rmsnorm=True, norm_before_gate=False, dt_rank="auto",
                 dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0, dt_init_floor=1e-4,
                 dt_limit=(0.0, float("inf")), dropout=0., conv_bias=True, bias=False, chunk_size=256,
                 use_mem_eff_path=True, layer_idx=None, process_group=None, sequence_parallel=True, device=None,
                 dtype=None, **kwargs):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        if process_group is not None:
            raise RuntimeError("tensor-parallel CrossMamba is dead code in the reference and is not built")
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.process_group, self.sequence_parallel, self.world_size, self.local_rank = None, sequence_parallel, 1, 0
        self.d_inner = int(self.expand * self.d_model)
        self.headdim = headdim
        self.d_ssm = self.d_inner if d_ssm is None else d_ssm
        self.ngroups = ngroups
        assert self.d_ssm % self.headdim == 0
        self.nheads = self.d_ssm // self.headdim
        self.D_has_hdim, self.rmsnorm, self.norm_before_gate = D_has_hdim, rmsnorm, norm_before_gate
        self.dt_limit, self.chunk_size, self.use_mem_eff_path, self.layer_idx = dt_limit, chunk_size, use_mem_eff_path, layer_idx
        self.dt_rank = math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank
        GN = self.ngroups * self.d_state

        self.in_proj = nn.Linear(self.d_model, 2 * self.d_inner + 2 * GN + self.nheads, bias=bias, **fk)   # unused in forward
        self.skip_in_proj = nn.Linear(self.d_model, 2 * self.d_inner - self.d_ssm, bias=bias, **fk)        # [z0, x0, z]
        self.xs_in_proj = nn.Linear(self.d_model, self.d_ssm, bias=bias, **fk)                             # scanned sequence
        self.BCdts_in_proj = nn.Linear(self.d_model, 2 * GN + self.nheads, bias=bias, **fk)                # [B, C, dt]
        conv = lambda c: nn.Conv2d(c, c, groups=c, bias=conv_bias, kernel_size=d_conv, padding=(d_conv - 1) // 2, **fk)
        self.conv2d = conv(self.d_ssm + 2 * GN + self.nheads)                                              # unused in forward
        self.xs_conv2d = conv(self.d_ssm)
        self.BCdts_conv2d = conv(2 * GN + self.nheads)
        self.act = nn.SiLU()
        dt = torch.exp(torch.rand(self.nheads, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = torch.clamp(dt, min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))
        self.dt_bias = nn.Parameter(torch.stack([inv_dt] * 4, dim=0))
        self.dt_bias._no_weight_decay = True
        self.A_logs = SS2D_with_SSD.A_log_init(A_init_range, self.nheads, dtype, copies=4)
        self.Ds = SS2D_with_SSD.D_init(self.d_ssm, self.D_has_hdim, self.nheads, copies=4)
        if self.rmsnorm:
            self.norm = RMSNormGated(self.d_ssm, eps=1e-5, norm_before_gate=self.norm_before_gate,
                                     group_size=self.d_ssm // ngroups, **fk)
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **fk)
        self.dropout = nn.Dropout(dropout) if dropout > 0. else None

    A_log_init = staticmethod(SS2D_with_SSD.A_log_init)
    D_init = staticmethod(SS2D_with_SSD.D_init)

    def forward(self, u1, u2, u2_cat_u1, u1_cat_u2, seq_idx=None, cu_seqlens=None):
        """u1, u2: the two modalities' (B,H,W,d_model) features; u2_cat_u1 parameterises the scan of u1 and u1_cat_u2 that
        of u2 (CrossMamba_fusion_2b2.py:235-237,263-268).  Returns (out1, out2), each (B,H,W,d_model)."""
        _lib.require_cuda(u1, u2, u2_cat_u1, u1_cat_u2)
        if seq_idx is not None or cu_seqlens is not None:
            raise RuntimeError("CrossMamba: seq_idx / cu_seqlens are never passed by the reference's models and are not built")

        def one(u, u_param):
            zx = proj(self.skip_in_proj, u)
            d_mlp = (zx.shape[-1] - self.d_ssm) // 2
            z0, x0, z = torch.split(zx, [d_mlp, d_mlp, self.d_ssm], dim=-1)
            xs = ssd_dwconv_silu(self.xs_conv2d, proj(self.xs_in_proj, u), self.d_conv)                 # (B,H,W,d_ssm)
            bcd = ssd_dwconv_silu(self.BCdts_conv2d, proj(self.BCdts_in_proj, u_param), self.d_conv)    # (B,H,W,2GN+nheads)
            return ssd_tail(self, ssd_scan_merge(self, torch.cat([xs, bcd], dim=-1)), z, z0, x0, d_mlp)

        return one(u1, u2_cat_u1), one(u2, u1_cat_u2)


# ---- the fusion network around the block: CrossMamba_fusion_2b2.py:650-1285 ------------------------------------------------
import torch.utils.checkpoint as checkpoint   # noqa: E402
from functools import partial                 # noqa: E402

from . import cnn_mamba as _cm                # noqa: E402
from . import medmamba as _mm                 # noqa: E402
from .medmamba import PatchEmbed2D as _PatchEmbedNCHW, PatchMerging2D   # noqa: E402,F401


class SS_Conv_SSD(_cm.SS_Conv_SSD):
    """CrossMamba_fusion_2b2.py:650-714: the two-branch block of CNN_Mamba.py with an explicit `input_dim` (the channel
    count that really enters forward; `hidden_dim` is only its default)."""

    def __init__(self, hidden_dim: int = 0, input_dim=None, drop_path: float = 0,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), attn_drop_rate: float = 0, d_state: int = 64, **kwargs):
        self.input_dim = hidden_dim if input_dim is None else input_dim
        super().__init__(hidden_dim=self.input_dim, drop_path=drop_path, norm_layer=norm_layer,
                         attn_drop_rate=attn_drop_rate, d_state=d_state, **kwargs)


class PatchEmbed2D(_PatchEmbedNCHW):
    """CrossMamba_fusion_2b2.py:719-743: same parameters as MedMamba's, but forward takes channel-LAST images (B,H,W,C)."""

    def forward(self, x):
        return super().forward(x.permute(0, 3, 1, 2))


def _conv1x1_last(conv, x):
    """`conv(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)` for a 1 x 1 / stride 1 `nn.Conv2d` on channel-last tokens (the two bridges and the
    output head, CrossMamba_fusion_2b2.py:1158-1160,1283-1285): a Linear over the last dimension, which is how it runs here.  Handed to
    the convolution library as a permuted view, the three of them took MIOpen's `naive_conv_*_nonpacked_*` / a grouped-convolution
    weight-gradient kernel at 175-204 ms PER CALL -- 59 % of the kernel time of the fusion step (profiles/r03_vfefm_kernels_before.txt)."""
    if not (type(conv) is nn.Conv2d and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.dilation == (1, 1) and conv.groups == 1 and x.shape[-1] == conv.in_channels):
        return conv(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    w = conv.weight.view(conv.out_channels, conv.in_channels)
    if conv.out_channels % 8 == 0 and conv.in_channels % 8 == 0 and x.is_cuda:
        from .ss2d_ops import linear_splitk            # this package's GEMMs (as cnn_mamba.proj); the bias joins as one broadcast add
        y = linear_splitk(x, w)
        return y if conv.bias is None else y + conv.bias.to(y.dtype)
    return torch.nn.functional.linear(x, w, conv.bias)


def _pixel_shuffle_last(x, p, c):
    """'b h w (p1 p2 c) -> b (h p1) (w p2) c' (the einops pattern of CrossMamba_fusion_2b2.py:804-808,828-829)."""
    B, H, W, _ = x.shape
    return x.view(B, H, W, p, p, c).permute(0, 1, 3, 2, 4, 5).reshape(B, H * p, W * p, c)


class PatchExpand2D(nn.Module):
    """CrossMamba_fusion_2b2.py:788-813: Linear(dim -> dim*scale), 2x2 pixel shuffle to dim/scale channels, LayerNorm."""

    def __init__(self, dim, dim_scale=2, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.dim_scale = dim, dim_scale
        self.expand = nn.Linear(dim, dim * dim_scale, bias=False)
        self.norm = norm_layer(dim // dim_scale)

    def forward(self, x):
        x = _pixel_shuffle_last(self.expand(x), self.dim_scale, self.dim // self.dim_scale)
        return _mm._norm_rows(self.norm, x.float().contiguous() if x.is_cuda else x, out_bf16=False)


class Final_PatchExpand2D(nn.Module):
    """CrossMamba_fusion_2b2.py:816-832: Linear(dim -> 4 dim), 4x4 pixel shuffle to dim/4 channels, LayerNorm."""

    def __init__(self, dim, dim_scale=4, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.dim_scale = dim, dim_scale
        self.expand = nn.Linear(self.dim, dim_scale * self.dim, bias=False)
        self.norm = norm_layer(self.dim // dim_scale)

    def forward(self, x):
        C = x.shape[-1]
        x = _pixel_shuffle_last(self.expand(x), self.dim_scale, C // self.dim_scale)
        return _mm._norm_rows(self.norm, x.float().contiguous() if x.is_cuda else x, out_bf16=False)


def _cross_context(layer, x1, x2):
    """What parameterises each modality's scan (CrossMamba_fusion_2b2.py:917-930,1048-1064): returns (u2_cat_u1, u1_cat_u2)."""
    if layer.cat_method == "add":
        s = x1 + x2
        return s, s
    if layer.cat_method == "stack":
        u = layer.cat_proj(torch.cat([x1, x2], dim=-1))
        return u, u
    if layer.cat_method == "cls" and isinstance(layer, upLayer):
        raise RuntimeError("cat_method='cls' is unfinished in the reference's upLayer (CrossMamba_fusion_2b2.py:1062-1063)")
    return x2, x1                                        # 'none' (and downLayer's fall-through)


def _run_blocks(blocks, x, use_checkpoint):
    for blk in blocks:
        x = checkpoint.checkpoint(blk, x, use_reentrant=False) if use_checkpoint else blk(x)
    return x


def _block_stack(dim, depth, drop_path, norm_layer, attn_drop, d_state):
    return nn.ModuleList([
        SS_Conv_SSD(hidden_dim=dim, input_dim=dim, drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                    norm_layer=norm_layer, attn_drop_rate=attn_drop, d_state=d_state) for i in range(depth)])


class downLayer(nn.Module):
    """Encoder stage (CrossMamba_fusion_2b2.py:836-946): `depth` SS_Conv_SSD blocks per modality, one CrossMamba fusion with
    residuals, the pre-fusion features pushed to `skip_list`, optional PatchMerging per modality."""

    def __init__(self, dim, depth, cat_method, attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None,
                 use_checkpoint=False, d_state=128, **kwargs):
        super().__init__()
        self.dim, self.use_checkpoint, self.cat_method = dim, use_checkpoint, cat_method
        self.cat_proj = None
        if cat_method == "stack":
            self.cat_proj = nn.Linear(dim * 2, dim)
        elif cat_method == "cls":
            self.cat_proj = nn.Linear(dim, dim)
        self.blocks1 = _block_stack(dim, depth, drop_path, norm_layer, attn_drop, d_state)
        self.blocks2 = _block_stack(dim, depth, drop_path, norm_layer, attn_drop, d_state)
        self.fusion = CrossMamba(d_model=dim, dropout=attn_drop)
        self.downsample1 = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None
        self.downsample2 = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x1, x2, skip_list=None):
        x1 = _run_blocks(self.blocks1, x1, self.use_checkpoint)
        x2 = _run_blocks(self.blocks2, x2, self.use_checkpoint)
        c21, c12 = _cross_context(self, x1, x2)
        f1, f2 = self.fusion(x1, x2, c21, c12)
        f1, f2 = x1 + f1, x2 + f2
        if skip_list is not None:
            skip_list.append((x1, x2))
        if self.downsample1 is not None:
            f1, f2 = self.downsample1(f1), self.downsample2(f2)
        return f1, f2


class upLayer(nn.Module):
    """Decoder stage (CrossMamba_fusion_2b2.py:949-1075): optional PatchExpand per modality, skip concat + Linear(2dim->dim),
    `depth` blocks per modality, CrossMamba fusion with residuals."""

    def __init__(self, dim, depth, cat_method, attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, upsample=None,
                 upsample_in_dim=None, skip=True, use_checkpoint=False, d_state=128, **kwargs):
        super().__init__()
        self.dim, self.use_checkpoint, self.cat_method = dim, use_checkpoint, cat_method
        self.cat_proj = None
        if cat_method == "stack":
            self.cat_proj = nn.Linear(dim * 2, dim)
        elif cat_method == "cls":
            self.cat_proj = nn.Linear(dim, dim)
        self.in_proj1 = nn.Linear(dim * 2, dim)
        self.in_proj2 = nn.Linear(dim * 2, dim)
        self.blocks1 = _block_stack(dim, depth, drop_path, norm_layer, attn_drop, d_state)
        self.blocks2 = _block_stack(dim, depth, drop_path, norm_layer, attn_drop, d_state)
        self.fusion = CrossMamba(d_model=dim, dropout=attn_drop)
        if upsample is not None:
            assert upsample_in_dim is not None, "upsample_in_dim must be provided when upsample is not None"
            self.upsample1 = upsample(dim=upsample_in_dim, norm_layer=norm_layer)
            self.upsample2 = upsample(dim=upsample_in_dim, norm_layer=norm_layer)
        else:
            self.upsample1 = self.upsample2 = None
        self.skip = skip

    def forward(self, x10, x20, x1_down, x2_down):
        if self.upsample1 is not None:
            x10, x20 = self.upsample1(x10), self.upsample2(x20)
        if self.skip:
            assert x10.shape[1:3] == x1_down.shape[1:3], f"Shape mismatch: x10={x10.shape}, x1_down={x1_down.shape}"
            assert x20.shape[1:3] == x2_down.shape[1:3], f"Shape mismatch: x20={x20.shape}, x2_down={x2_down.shape}"
            x1 = self.in_proj1(torch.cat([x10, x1_down.to(x10.dtype)], dim=-1))
            x2 = self.in_proj2(torch.cat([x20, x2_down.to(x20.dtype)], dim=-1))
        else:
            x1, x2 = x10, x20
        assert x1.shape[-1] == self.dim and x2.shape[-1] == self.dim, \
            f"upLayer forward channel mismatch: expect dim={self.dim}, got x1={x1.shape[-1]}, x2={x2.shape[-1]}"
        x1 = _run_blocks(self.blocks1, x1.float(), self.use_checkpoint)
        x2 = _run_blocks(self.blocks2, x2.float(), self.use_checkpoint)
        c21, c12 = _cross_context(self, x1, x2)
        f1, f2 = self.fusion(x1, x2, c21, c12)
        return x1 + f1, x2 + f2


class VFEFM(nn.Module):
    """Two-modality fusion U-Net (CrossMamba_fusion_2b2.py:1078-1285): per-modality PatchEmbed, 4 encoder stages
    (`downLayer`), 1x1 bridge convs, 4 decoder stages (`upLayer`, skips matched by spatial size), LayerNorm over the two
    decoded streams, Linear(2C->C), 4x patch expand, 1x1 conv to ONE channel.  forward(x1, x2: (B,in_chans,H,W)) ->
    (B,1,H,W).  Same constructor arguments and state_dict keys as the reference class."""

    def __init__(self, patch_size=4, in_chans=3, num_classes=1000, depths=[2, 2, 4, 2], dims=[128, 256, 512, 1024],
                 depths_decoder=[2, 9, 2, 2], dims_decoder=[1024, 512, 256, 128], d_state=128, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, patch_norm=True, use_checkpoint=False,
                 cat_method="stack", **kwargs):
        super().__init__()
        self.ape = False
        self.pos_drop1, self.pos_drop2 = nn.Dropout(p=drop_rate), nn.Dropout(p=drop_rate)
        self.norm = nn.LayerNorm(dims_decoder[-1] * 2)
        self.num_layers, self.embed_dim, self.num_features, self.dims = len(depths), dims[0], dims[-1], dims
        pe = lambda: PatchEmbed2D(patch_size=patch_size, in_chans=in_chans, embed_dim=self.embed_dim,
                                  norm_layer=norm_layer if patch_norm else None)
        self.patch_embed1, self.patch_embed2 = pe(), pe()
        ds = math.ceil(dims[0] / 6) if d_state is None else d_state
        dpr = [r.item() for r in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i, depth in enumerate(depths):
            self.layers.append(downLayer(
                dim=dims[i], depth=depth, cat_method=cat_method, d_state=ds, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging2D if i < self.num_layers - 1 else None, use_checkpoint=use_checkpoint))
        self.num_layers_up, self.final_dim, self.dims_decoder = len(depths_decoder), dims_decoder[-1], dims_decoder
        self.final_cat_proj = nn.Linear(self.final_dim * 2, self.final_dim)
        self.final_expand = Final_PatchExpand2D(dim=dims_decoder[-1])
        self.final_conv = nn.Conv2d(dims_decoder[-1] // 4, 1, 1)
        dpr = [r.item() for r in torch.linspace(0, drop_path_rate, sum(depths_decoder))]
        self.layers_up = nn.ModuleList()
        for i, depth in enumerate(depths_decoder):
            last = i == self.num_layers_up - 1
            self.layers_up.append(upLayer(
                dim=dims_decoder[i] if last else dims_decoder[i] // 2, depth=depth, cat_method=cat_method, d_state=ds,
                drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[sum(depths_decoder[:i]):sum(depths_decoder[:i + 1])],
                norm_layer=norm_layer, upsample=None if last else PatchExpand2D,
                upsample_in_dim=None if last else dims_decoder[i], skip=i != 0, use_checkpoint=use_checkpoint))
        self.bridge1 = nn.Conv2d(dims[-1], dims_decoder[0], 1)
        self.bridge2 = nn.Conv2d(dims[-1], dims_decoder[0], 1)
        self.apply(self._init_weights)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _init_weights(self, m: nn.Module):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"absolute_pos_embed"}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {"relative_position_bias_table"}

    def forward_down(self, x1, x2):
        """x1, x2 channel-last images (B,H,W,in_chans) -> deepest features + the per-stage skips."""
        x1 = self.pos_drop1(self.patch_embed1(x1))
        x2 = self.pos_drop2(self.patch_embed2(x2))
        skip = []
        for layer in self.layers:
            x1, x2 = layer(x1, x2, skip_list=skip)
        return x1, x2, skip

    def forward_up(self, x1, x2, skip):
        x1 = _conv1x1_last(self.bridge1, x1)
        x2 = _conv1x1_last(self.bridge2, x2)
        skip_rev = list(reversed(skip))
        for j, layer_up in enumerate(self.layers_up):
            if j == 0:
                u1, u2 = x1, x2
            else:
                H, W = x1.shape[1], x1.shape[2]
                target = (H * 2, W * 2) if layer_up.upsample1 is not None else (H, W)
                u1 = u2 = None
                for s1, s2 in skip_rev:                   # first skip (deepest first) with the stage's spatial size
                    if tuple(s1.shape[1:3]) == target:
                        u1, u2 = s1, s2
                        break
                assert u1 is not None, f"No skip with spatial size {target} found!"
            x1, x2 = layer_up(x1, x2, u1, u2)
        x = self.norm(torch.cat([x1, x2], dim=-1))
        return self.final_expand(self.final_cat_proj(x))

    def forward(self, x1, x2):
        _lib.require_cuda(x1, x2)
        x1, x2, skip = self.forward_down(x1.permute(0, 2, 3, 1), x2.permute(0, 2, 3, 1))
        x = self.forward_up(x1, x2, skip)
        return _conv1x1_last(self.final_conv, x).permute(0, 3, 1, 2)
