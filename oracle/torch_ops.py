"""TEST INFRASTRUCTURE (like everything under oracle/): the reference's query path restated with the SAME torch CPU
operators it calls, for one purpose -- timing "the reference CPU recon path" on the host cores of a box that has no
/root/reference (bench.py's cpu_baseline leg, `cpu_baseline_reference_ops`), and being checked against the
reference-generated goldens on the way (tests/test_oracle_golden.py::test_torch_ops_query_matches_reference).

The C oracle (oracle/c) restates the ARITHMETIC and is several times faster than the reference's own CPU path (hand
tiling, OpenMP over points); this module restates the OPERATOR SEQUENCE, so that its time is the reference's time:

  orthogonal       torch.baddbmm(trans, rot, points)                       geometry.py:27-29
  in-image mask    four comparisons on x and y                             MonoPortNet.py:74
  z feature        z * scale                                               DepthNormalizer.py:32
  index            F.grid_sample(feat, uv[B,N,1,2], align_corners=True)    geometry.py:11-16
  concat           torch.cat([sampled..., z_feat], 1)                      MonoPortNet.py:82-83
  MLP              Conv1d(k=1) on cat([y, x]) + F.leaky_relu, last op      SurfaceClassifier.py:47-69
  mask multiply    in_img[:, None].float() * pred                          MonoPortNet.py:89

Inputs are the oracle's plain arrays (feature map [C,H,W], points [3,N], calib [4,4] or [3,4], layers = [(W[out,in],
b[out])]); nothing here is imported by the product.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


@torch.no_grad()
def query(feat, points, calib, layers, last_op, z_scale):
    """[Cout, N] f32 numpy: the reference's eval-mode MonoPortNet.query on one stage / one level of features."""
    feat = _t(feat)[None]                     # [1,C,H,W]
    pts = _t(points)[None]                    # [1,3,N]
    cal = _t(calib).reshape(1, -1, 4)         # [1,4,4] or [1,3,4]
    rot, trans = cal[:, :3, :3], cal[:, :3, 3:4]
    xyz = torch.baddbmm(trans, rot, pts)
    xy, z = xyz[:, :2, :], xyz[:, 2:3, :]
    in_img = (xy[:, 0] >= -1.0) & (xy[:, 0] <= 1.0) & (xy[:, 1] >= -1.0) & (xy[:, 1] <= 1.0)
    z_feat = z * z_scale
    uv = xy.transpose(1, 2).unsqueeze(2)
    sampled = F.grid_sample(feat, uv, align_corners=True)[:, :, :, 0]
    x = torch.cat([sampled, z_feat], 1)
    y = x
    n = len(layers)
    for i, (w, b) in enumerate(layers):
        inp = y if i == 0 else torch.cat([y, x], 1)
        y = F.conv1d(inp, _t(w)[:, :, None], _t(b))
        if i != n - 1:
            y = F.leaky_relu(y)
    if last_op == 1:
        y = torch.sigmoid(y)
    elif last_op == 2:
        y = torch.tanh(y)
    out = in_img[:, None].float() * y
    return out[0].numpy()
