"""FusionMamba's stride-2 scan / merge permutations against vectors produced by running the reference's
cross.py (tools/make_golden_cross.py): bit-exact, forward and backward, including odd sizes (zero padding) and 1x1."""
import glob
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "effscan_perm_*.npz"))), ids=os.path.basename)
def test_efficient_scan_and_merge_are_the_reference_permutations(path):
    from medical_image_classification_amd.efficient_scan import EfficientMerge, EfficientScan
    g = np.load(path)
    x = torch.from_numpy(g["x"]).requires_grad_()
    xs = EfficientScan.apply(x, 2)
    xs.backward(torch.from_numpy(g["g_xs"]))
    assert np.array_equal(xs.detach().numpy(), g["xs"]) and np.array_equal(x.grad.numpy(), g["dx"])
    ys = torch.from_numpy(g["ys"]).requires_grad_()
    H, W = g["x"].shape[2:]
    y = EfficientMerge.apply(ys, H, W, 2)
    y.backward(torch.from_numpy(g["g_y"]))
    assert np.array_equal(y.detach().numpy(), g["y"]) and np.array_equal(ys.grad.numpy(), g["dys"])


def test_merge_inverts_scan_and_step_size_is_checked():
    from medical_image_classification_amd.efficient_scan import EfficientMerge, EfficientScan
    x = torch.randn(2, 3, 9, 5)
    assert torch.equal(EfficientMerge.apply(EfficientScan.apply(x, 2), 9, 5, 2).view_as(x), x)
    with pytest.raises(RuntimeError):
        EfficientScan.apply(x, 3)
