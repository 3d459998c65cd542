"""Depth feature normaliser (mirror of monoport/lib/modeling/normalizers/DepthNormalizer.py)."""
import torch.nn as nn


class DepthNormalizer(nn.Module):
    """z_feat = z * scale (DepthNormalizer.py:32).  The soft one-hot branch (:17-30) is disabled
    in every PIFu config (config.py:42) and not implemented.  Inside ``MonoPortNet.query`` the
    multiply is folded into the fused HIP kernel; ``forward`` serves stand-alone callers."""

    def __init__(self, scale=512 // 2 / 200.0, soft_onehot=False):
        super().__init__()
        if soft_onehot:
            raise NotImplementedError("soft_onehot depth features are not part of the PIFu configs")
        self.scale = float(scale)

    def forward(self, z, calibs=None, index_feat=None):
        return z * self.scale


def PIFuNomalizer(*args, **kwargs):  # (sic) -- the reference's spelling, DepthNormalizer.py:36
    return DepthNormalizer(scale=512 // 2 / 200.0)
