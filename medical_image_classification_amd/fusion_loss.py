"""Fusion training loss of the reference (CrossMamba/FusionMamba/loss.py:163-205, `Fusionloss`): stock torch ops, no kernels of
this package -- SURVEY.md 8f-2 keeps it that way.  total = 10 * (1 - MS-SSIM terms) + 10 * MSE(max(vis, ir), fused) +
1 * L1(max(|sobel vis|, |sobel ir|), |sobel fused|), on single-channel images clamped to [0, 1].

Restated, not imported: the reference module drags in `.cuda()` at construction (loss.py:155-156).  One simplification that
does not change any value: every image reaching SSIM has been clamped to [0, 1], so the reference's data-dependent dynamic
range probe (loss.py:35-45, a host sync per call) always yields L = 1; it is a constant here."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

_MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _gauss_band(size, n, device, dtype, sigma=1.5):
    """(n - size + 1, n) banded matrix whose rows hold the normalised 1-D Gaussian of loss.py:16-18: the reference's 2-D
    window is the outer product of that vector with itself (loss.py:22-26), so its valid (unpadded) depthwise conv2d equals
    G_H @ x @ G_W^T exactly.  Written as two GEMMs on purpose: the single-channel 11x11 convolutions of the 5-level chain
    take MIOpen down with a memory access fault in backward at 224x224 (ROCm 7.2, MI355X, seen in this repo's round 1)."""
    g = torch.tensor([math.exp(-((i - size // 2) ** 2) / (2.0 * sigma * sigma)) for i in range(size)], dtype=torch.float64)
    g = (g / g.sum()).to(dtype)
    m = torch.zeros(n - size + 1, n, dtype=dtype)
    for r in range(n - size + 1):
        m[r, r:r + size] = g
    return m.to(device)


def _ssim_and_cs(a, b, window_size=11):
    """Mean SSIM and mean contrast-structure term of two (B,C,H,W) images in [0,1] (loss.py:32-80, no padding)."""
    H, W = a.shape[2:]
    k = min(window_size, H, W)
    gh, gw = _gauss_band(k, H, a.device, a.dtype), _gauss_band(k, W, a.device, a.dtype).t()
    blur = lambda t: torch.matmul(torch.matmul(gh, t), gw)
    mu_a, mu_b = blur(a), blur(b)
    var_a, var_b, cov = blur(a * a) - mu_a * mu_a, blur(b * b) - mu_b * mu_b, blur(a * b) - mu_a * mu_b
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    v1, v2 = 2.0 * cov + c2, var_a + var_b + c2
    cs = (v1 / v2).mean()
    ssim = (((2.0 * mu_a * mu_b + c1) * v1) / ((mu_a * mu_a + mu_b * mu_b + c1) * v2)).mean()
    return ssim, cs


def ms_ssim(a, b, window_size=11, normalize=True):
    """5-level MS-SSIM exactly as loss.py:83-108 combines it (prod over levels of cs_l^w_l * ssim_last^w_l, l < 4)."""
    w = torch.tensor(_MS_WEIGHTS, device=a.device, dtype=a.dtype)
    sims, css = [], []
    for _ in range(len(_MS_WEIGHTS)):
        s, c = _ssim_and_cs(a, b, window_size)
        sims.append(s); css.append(c)
        a, b = F.avg_pool2d(a, 2), F.avg_pool2d(b, 2)
    sims, css = torch.stack(sims), torch.stack(css)
    if normalize:
        sims, css = (sims + 1) / 2, (css + 1) / 2
    return torch.prod((css ** w)[:-1] * (sims ** w)[-1])


class FusionLoss(nn.Module):
    def __init__(self):
        super().__init__()

    @staticmethod
    def _sobel(x):
        """|x (*) kx| + |x (*) ky| with zero padding 1 (loss.py:143-160), kx = [[-1,0,1],[-2,0,2],[-1,0,1]], ky = [[1,2,1],[0,0,0],
        [-1,-2,-1]], as shifted differences."""
        p = F.pad(x, (1, 1, 1, 1))
        t, m, b = p[..., :-2, :], p[..., 1:-1, :], p[..., 2:, :]
        gx = (t[..., 2:] - t[..., :-2]) + 2.0 * (m[..., 2:] - m[..., :-2]) + (b[..., 2:] - b[..., :-2])
        gy = (t[..., :-2] + 2.0 * t[..., 1:-1] + t[..., 2:]) - (b[..., :-2] + 2.0 * b[..., 1:-1] + b[..., 2:])
        return gx.abs() + gy.abs()

    def forward(self, image_vis, image_ir, generate_img):
        """-> (total, loss_in, ssim_value, loss_grad), the tuple the reference loop logs (train.py:131-152)."""
        y = image_vis[:, :1].clamp(0, 1)
        ir = image_ir[:, :1].clamp(0, 1)
        if generate_img.size(1) == 3:
            generate_img = 0.299 * generate_img[:, 0:1] + 0.587 * generate_img[:, 1:2] + 0.114 * generate_img[:, 2:3]
        gen = generate_img.clamp(0, 1)
        ssim_value = 0.5 * (1 - ms_ssim(gen, y)) + 0.5 * (1 - ms_ssim(gen, ir))
        loss_in = F.mse_loss(torch.max(y, ir), gen)
        loss_grad = F.l1_loss(torch.max(self._sobel(y), self._sobel(ir)), self._sobel(gen))
        return 10 * ssim_value + 10 * loss_in + loss_grad, loss_in, ssim_value, loss_grad
