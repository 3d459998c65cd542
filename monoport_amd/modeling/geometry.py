"""Per-point geometry ops of the PIFu query (mirror of monoport/lib/modeling/geometry.py).

``index`` and ``orthogonal`` run as HIP kernels (monoport_amd/csrc/query.hip) -- inside
``MonoPortNet.query`` they are fused with the MLP and never launched on their own; the
stand-alone entry points exist for callers such as colorization (RTL/main.py:237).
"""
import torch

from .. import ops


def index(feat, uv):
    """Bilinear sample (grid_sample, align_corners=True, zero padding -- geometry.py:4-16).

    feat [1,C,H,W] (NCHW, as the encoders emit) or an already packed channels-last [H,W,C] map;
    uv [1,2,N] in [-1,1].  Returns [1,C,N].
    """
    if feat.dim() == 4:
        if feat.shape[0] != 1:
            return torch.cat([index(feat[b:b + 1], uv[b:b + 1]) for b in range(feat.shape[0])], 0)
        feat = ops.pack_features(feat)
    return ops.index(feat, uv)


def orthogonal(points, calibrations, transforms=None):
    """xyz = R p + t with R = calib[:, :3, :3], t = calib[:, :3, 3:4] (geometry.py:19-34).

    points [B,3,N]; calibrations [B,>=3,4].  ``transforms`` (the training-time image-space affine,
    geometry.py:30-33) is never passed on the inference path (MonoPortNet.py:69,
    RTL/main.py:179-182) and is not supported.
    """
    if transforms is not None:
        raise NotImplementedError("orthogonal(transforms=...) is outside the reconstruction path")
    if points.shape[0] != 1:
        return torch.cat([orthogonal(points[b:b + 1], calibrations[b:b + 1])
                          for b in range(points.shape[0])], 0)
    return ops.orthogonal(points, calibrations)


def perspective(points, calibrations, transforms=None):
    """Pinhole projection (geometry.py:37-55).  No PIFu config selects it (config.py:33,:59);
    kept importable for API parity, computed with stock tensor ops."""
    if transforms is not None:
        raise NotImplementedError("perspective(transforms=...) is outside the reconstruction path")
    cam = calibrations[:, :3, :3] @ points + calibrations[:, :3, 3:4]
    depth = cam[:, 2:3, :]
    return torch.cat([cam[:, :2, :] / depth, depth], 1)
