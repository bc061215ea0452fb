"""MI355X-native (gfx950) implementation of the SS2D / selective-scan hot path of
leaf-down/Medical_image_classification (MedMamba).  See DESIGN.md.

Public surface (same names as the reference):
  selective_scan_fn, SelectiveScanFn                       (mamba_ssm/ops/selective_scan_interface.py)
  SS2D, SS_Conv_SSM, VSSLayer, VSSM, PatchEmbed2D, PatchMerging2D, channel_shuffle   (MedMamba.py)
  aliases: VSSBlock = SS_Conv_SSM, MedMamba = VSSM
  medical_image_classification_amd.cross: SS2D, SS2D_cross_new, VSSBlock_new, VSSBlock_Cross_new, ...  (FusionMamba cross.py)
"""
from .selective_scan_interface import SelectiveScanFn, selective_scan_fn  # noqa: F401

__all__ = ["SelectiveScanFn", "selective_scan_fn"]


def __getattr__(name):  # lazy: the model surface pulls in einops-free torch modules only when asked for
    if name in ("SS2D", "SS_Conv_SSM", "VSSLayer", "VSSM", "PatchEmbed2D", "PatchMerging2D", "channel_shuffle",
                "VSSBlock", "MedMamba", "DropPath"):
        from . import medmamba
        return getattr(medmamba, name)
    raise AttributeError(name)
