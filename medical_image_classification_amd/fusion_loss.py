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


def _gauss_window(size, channels, device, dtype, sigma=1.5):
    g = torch.tensor([math.exp(-((i - size // 2) ** 2) / (2.0 * sigma * sigma)) for i in range(size)])
    g = g / g.sum()
    return torch.outer(g, g).to(device=device, dtype=dtype).expand(channels, 1, size, size).contiguous()


def _ssim_and_cs(a, b, window_size=11):
    """Mean SSIM and mean contrast-structure term of two (B,C,H,W) images in [0,1] (loss.py:32-80, no padding)."""
    C, H, W = a.shape[1:]
    win = _gauss_window(min(window_size, H, W), C, a.device, a.dtype)
    blur = lambda t: F.conv2d(t, win, groups=C)
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
        kx = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]])
        self.register_buffer("kx", kx.view(1, 1, 3, 3), persistent=False)
        self.register_buffer("ky", (-kx.t()).contiguous().view(1, 1, 3, 3), persistent=False)

    def _sobel(self, x):
        return F.conv2d(x, self.kx, padding=1).abs() + F.conv2d(x, self.ky, padding=1).abs()

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
