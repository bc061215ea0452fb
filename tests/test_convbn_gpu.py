"""The conv branch's BatchNorms folded into the 3x3 convolutions (csrc/conv3x3.hip `BNIN` / `STATS` forward, `BRED` backward,
ms_bn_apply_sums_nhwc / ms_bn_bwd_apply_sums_nhwc; host: block_ops._ConvBnConvBn) against (a) the unfused kernel sequence of the same package (statistics / finalize / apply launches), which the
existing tests hold to float64 torch, and (b) float64 torch directly for the statistics the epilogue accumulates.
Reference: MedMamba.py:517-527 (`conv33conv33conv11`), applied at :533-535.

Tolerances.  Both paths compute in bf16 with fp32 statistics; they differ in HOW the statistics are summed (pivoted sums through
atomics vs per-workgroup Chan merges), i.e. by ~1e-6 relative in mean / rstd, which moves individual bf16 activations by one ulp
(2^-8 relative) here and there.  Outputs are therefore held to 2e-2 of the tensor's max-norm element-wise and 4e-3 in relative L2;
gradients to 2e-2 relative L2; batch statistics / running statistics to 1e-4 relative (+1e-5 absolute)."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _branch(C):
    return nn.Sequential(nn.BatchNorm2d(C), nn.Conv2d(C, C, 3, 1, 1), nn.BatchNorm2d(C), nn.ReLU(), nn.Conv2d(C, C, 3, 1, 1), nn.BatchNorm2d(C), nn.ReLU(),
                         nn.Conv2d(C, C, 1, 1), nn.ReLU())


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("B,C,H,W", [(4, 48, 56, 56), (3, 96, 28, 28), (2, 192, 14, 14), (5, 384, 7, 7), (2, 64, 24, 40), (1, 48, 9, 21)])
def test_conv_branch_fused_bn_matches_sequential(B, C, H, W):
    from medical_image_classification_amd import block_ops
    dev = torch.device("cuda:0")
    torch.manual_seed(C + H)
    seq_a = _branch(C).to(dev).train()
    with torch.no_grad():
        for m in seq_a:
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.5, 0.5)
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 2.0)
            if isinstance(m, nn.Conv2d):
                m.bias.uniform_(-1.0, 1.0)
    seq_b = copy.deepcopy(seq_a)
    x = (torch.randn(B, C, H, W, device=dev) * 1.5 + 0.7).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    outs, calls0 = [], block_ops.FOLD_CALLS
    for seq, fold in ((seq_a, True), (seq_b, False)):
        old = block_ops._BN_FOLD
        block_ops._BN_FOLD = fold
        try:
            xi = x.clone().requires_grad_()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = block_ops.conv_branch(seq, xi)
            y.float().backward(gy)
            torch.cuda.synchronize()
        finally:
            block_ops._BN_FOLD = old
        outs.append((y.detach().float(), xi.grad.detach().float(), {k: p.grad.detach().float() for k, p in seq.named_parameters()},
                     {k: b.detach().clone() for k, b in seq.named_buffers()}))
    assert block_ops.FOLD_CALLS == calls0 + 1            # the folded form ran exactly once (the first pass)
    (ya, dxa, ga, ba), (yb, dxb, gb, bb) = outs
    assert float((ya - yb).abs().max()) <= 2e-2 * float(yb.abs().max()) and _rel(ya, yb) <= 4e-3, (_rel(ya, yb),)
    assert _rel(dxa, dxb) <= 2e-2, ("dx", _rel(dxa, dxb))
    for k in gb:
        if k in ("1.bias", "4.bias"):                       # conv biases in front of a training-mode BatchNorm: exact zeros on both paths
            assert float(ga[k].abs().max()) == 0.0 and float(gb[k].abs().max()) == 0.0
            continue
        assert _rel(ga[k], gb[k]) <= 2e-2, (k, _rel(ga[k], gb[k]))
    for k in bb:
        if k.endswith("num_batches_tracked"):
            assert int(ba[k]) == int(bb[k]) == 1, k
        else:
            assert torch.allclose(ba[k], bb[k], rtol=1e-4, atol=1e-5), (k, float((ba[k] - bb[k]).abs().max()))


def test_conv_epilogue_statistics_vs_float64():
    """ms_conv3x3_bn_nhwc_bf16 (producer) + ms_bn_apply_sums_nhwc (consumer) on their own: batch mean / rstd of the stored bf16 outputs vs a
    float64 evaluation of the SAME tensor (1e-5 relative), with a running mean far from the batch mean as the pivot (a reset BatchNorm:
    running_mean 0, the convolution's output mean ~ 3 standard deviations away), ragged tiles, and the running-statistics update."""
    import ctypes
    from medical_image_classification_amd import _lib, block_ops, shadow
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    B, C, H, W = 3, 96, 13, 21
    conv = nn.Conv2d(C, C, 3, 1, 1).to(dev)
    bn = nn.BatchNorm2d(C).to(dev).train()
    with torch.no_grad():
        conv.weight.add_(0.02)                           # a common-sign component: channel means well away from zero
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
    x = (torch.rand(B, C, H, W, device=dev) + 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wb = shadow.bf16(conv.weight, conv=True)
    nf = (2 * _lib.BN_REPLICAS + 1) * C
    sums = torch.zeros(nf, device=dev)
    save = torch.empty(2, C, device=dev)
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    d = block_ops._bn_fold_desc(sums, bn.weight.detach(), bn.bias.detach(), conv.bias.detach(), bn, save)
    lib = _lib.lib()
    st = _lib.current_stream_ptr(dev)
    with _lib.on_device(dev):
        _lib.check(lib.ms_conv3x3_bn_nhwc_bf16(x.data_ptr(), wb.data_ptr(), y.data_ptr(), B, H, W, C, C, None, None, ctypes.byref(d), st), "conv")
        _lib.check(lib.ms_bn_apply_sums_nhwc(y.data_ptr(), ctypes.byref(d), 1, z.data_ptr(), B * H * W, C, st), "apply")
    torch.cuda.synchronize()
    y64 = y.double()
    mean = y64.mean(dim=(0, 2, 3)); var = y64.var(dim=(0, 2, 3), unbiased=False)
    assert float((mean.abs() / var.sqrt()).max()) > 2.0          # the case is what the docstring says
    rstd = (var + bn.eps).rsqrt()
    assert torch.allclose(save[0].double(), mean, rtol=1e-5, atol=1e-6), float((save[0].double() - mean).abs().max())
    assert torch.allclose(save[1].double(), rstd, rtol=2e-5), float((save[1].double() / rstd - 1).abs().max())
    want = torch.relu((y64 - mean[None, :, None, None]) * (rstd * bn.weight.double())[None, :, None, None] + bn.bias.double()[None, :, None, None])
    assert float((z.double() - want.detach()).abs().max()) <= 2 ** -7 * float(want.detach().abs().max())
    n = B * H * W
    assert torch.allclose(bn.running_mean.double(), 0.9 * rm0.double() + 0.1 * (mean + conv.bias.double()), rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.double(), 0.9 * rv0.double() + 0.1 * var * n / (n - 1), rtol=1e-4)
    assert int(bn.num_batches_tracked) == 1
    # error returns
    assert lib.ms_conv3x3_bn_nhwc_bf16(x.data_ptr(), wb.data_ptr(), y.data_ptr(), B, H, W, C, C, None, None, ctypes.byref(_lib.MsBnFold()), st) == -1
    assert lib.ms_bn_apply_sums_nhwc(None, ctypes.byref(d), 1, z.data_ptr(), B * H * W, C, st) == -1
