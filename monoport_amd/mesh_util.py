"""OBJ export of the marching-cubes mesh in the reference's on-disk format
(monoport/lib/mesh_util.py:223-242: ``v x y z [r g b]`` with %.4f, 1-based ``f i j k``), plus the
per-vertex colour query of BASELINE configs[2]."""
import numpy as np
import torch

from . import ops


def _as_numpy(a):
    return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)


def save_obj_mesh(mesh_path, verts, faces):
    """mesh_util.py:223-230."""
    v, f = _as_numpy(verts), _as_numpy(faces).astype(np.int64) + 1
    with open(mesh_path, "w") as fh:
        fh.write("".join("v %.4f %.4f %.4f\n" % tuple(row) for row in v))
        fh.write("".join("f %d %d %d\n" % tuple(row) for row in f))


def save_obj_mesh_with_color(mesh_path, verts, faces, colors):
    """mesh_util.py:233-242."""
    v, c = _as_numpy(verts), _as_numpy(colors)
    f = _as_numpy(faces).astype(np.int64) + 1
    with open(mesh_path, "w") as fh:
        fh.write("".join("v %.4f %.4f %.4f %.4f %.4f %.4f\n" % (tuple(a) + tuple(b))
                         for a, b in zip(v, c)))
        fh.write("".join("f %d %d %d\n" % tuple(row) for row in f))


@torch.no_grad()
def vertex_colors(netC, feat_tensor_C, verts, calib_tensor):
    """RGB in [0,1] for world-space vertices [V,3]: netC.query(...)*0.5+0.5 as RTL/main.py:239-244
    does for the visible-surface vertices."""
    pts = verts.t().contiguous()[None]  # [1,3,V]
    preds = netC.query(feat_tensor_C, points=pts, calibs=calib_tensor)[0]
    return (preds[0] * 0.5 + 0.5).t().contiguous()
