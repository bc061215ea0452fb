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
from .cnn_mamba import RMSNormGated, SS2D_with_SSD, ssd_dwconv_silu, ssd_scan_merge, ssd_tail

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
            zx = self.skip_in_proj(u)
            d_mlp = (zx.shape[-1] - self.d_ssm) // 2
            z0, x0, z = torch.split(zx, [d_mlp, d_mlp, self.d_ssm], dim=-1)
            xs = ssd_dwconv_silu(self.xs_conv2d, self.xs_in_proj(u), self.d_conv)                 # (B,H,W,d_ssm)
            bcd = ssd_dwconv_silu(self.BCdts_conv2d, self.BCdts_in_proj(u_param), self.d_conv)    # (B,H,W,2GN+nheads)
            return ssd_tail(self, ssd_scan_merge(self, torch.cat([xs, bcd], dim=-1)), z, z0, x0, d_mlp)

        return one(u1, u2_cat_u1), one(u2, u1_cat_u2)
