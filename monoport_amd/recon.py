"""Drop-ins for RTL/recon.py (``pifu_calib``, ``forward_vertices``) and for the colorization
closure of RTL/main.py:201-249, on top of the HIP kernels in csrc/vertices.hip."""
import numpy as np
import torch

from . import ops
from .modeling.MonoPortNet import capture_query

_FLIP_Y = np.diag([1.0, -1.0, 1.0, 1.0])  # RTL/recon.py:6-11


@torch.no_grad()
def pifu_calib(extrinsic, intrinsic, device="cuda:0"):
    """Calibration tensor [1,4,4] f32 = inv(K' E' diag(1,-1,1,1)) with the orthographic tweaks
    K'[2,2]=K[0,0], K'[2,3]=0, E'[2,3]=0 (RTL/recon.py:4-25).  Host-side float64 numpy; inputs are
    not modified."""
    k = np.array(intrinsic, copy=True)
    k[2, 2] = k[0, 0]
    k[2, 3] = 0
    e = np.array(extrinsic, copy=True)
    e[2, 3] = 0
    calib = np.linalg.inv(k @ e @ _FLIP_Y)
    return torch.from_numpy(calib).unsqueeze(0).float().to(device)


@torch.no_grad()
def forward_vertices(sdf, direction="front"):
    """Visible-surface vertices of an occupancy volume [1,1,D,H,W] (RTL/recon.py:27-89):
    X, Y int64 [N], Z f32 [N] (sub-voxel depth), norm f32 [N,3]; four Nones for ``sdf is None``.
    One host sync (N), like the reference's ``nonzero``."""
    if sdf is None:
        return None, None, None, None
    x, y, z, n, count = ops.forward_vertices_raw(sdf, direction)
    c = int(count.item())
    return x[:c], y[:c], z[:c], n[:c]


@torch.no_grad()
def forward_vertices_many(sdfs, direction="front"):
    """``[forward_vertices(s, direction) for s in sdfs]`` with ONE host sync for all the vertex
    counts (monoport_amd extension; the hook of a coalescing stage, stage_pipeline.Coalesced)."""
    idx = [i for i, s in enumerate(sdfs) if s is not None]
    raws = [None] * len(sdfs)
    if idx and len({tuple(sdfs[i].shape[-3:]) for i in idx}) == 1:  # one size: one set of launches for all of them
        for i, r in zip(idx, ops.forward_vertices_raw_batch([sdfs[i] for i in idx], direction)):
            raws[i] = r
    else:
        for i in idx:
            raws[i] = ops.forward_vertices_raw(sdfs[i], direction)
    live = [r for r in raws if r is not None]
    counts = torch.cat([r[4] for r in live]).cpu().tolist() if live else []
    out, k = [], 0
    for r in raws:
        if r is None:
            out.append((None, None, None, None))
        else:
            c = int(counts[k])
            k += 1
            out.append((r[0][:c], r[1][:c], r[2][:c], r[3][:c]))
    return out


def color_matrix(b_min, b_max, resolution):
    """voxel -> world matrix of RTL/main.py:204-210."""
    mat = np.eye(4, dtype=np.float32)
    length = np.asarray(b_max, np.float32).reshape(3) - np.asarray(b_min, np.float32).reshape(3)
    for i in range(3):
        mat[i, i] = length[i] / np.float32(resolution)
    mat[0:3, 3] = np.asarray(b_min, np.float32).reshape(3)
    return mat


@torch.no_grad()
def colorization(netC, feat_tensor_C, X, Y, Z, calib_tensor, norm=None, resolution=257,
                 mat_color=None):
    """[res,res,3] f32 render (RTL/main.py:212-249): normals as colour when ``norm`` is given,
    else netC.query on the vertices mapped to world space; ``None`` passes through (:214-215)."""
    if X is None:
        return None
    count = torch.tensor([X.shape[0]], dtype=torch.int32, device=X.device)
    if norm is not None:
        return ops.paint(X, Y, norm, 0, count, resolution, 0.5, 0.5, 0.0, 1.0)
    if mat_color is None:
        mat_color = color_matrix([-1, -1, -1], [1, 1, 1], resolution)
    if torch.is_tensor(mat_color):
        mat_color = mat_color.detach().cpu().numpy()
    device = calib_tensor.device
    feat_tensor_C = [[f.to(device) for f in feats] for feats in feat_tensor_C]  # main.py:229-230
    X, Y, Z = X.to(device), Y.to(device), Z.to(device)
    pts = ops.vertex_points(X, Y, Z.float(), count.to(device), resolution, mat_color)
    binding = netC.bind(feat_tensor_C, calib_tensor)
    preds = ops.query_counted(binding.mlp, binding.feat_hwc, pts, count.to(device), binding.calib,
                              binding.z_scale)
    return ops.paint(X, Y, preds, 1, count.to(device), resolution, 0.5, 0.5, -np.inf, np.inf)


@torch.no_grad()
def marching_cubes(sdf, level=0.5, b_min=(-1, -1, -1), b_max=(1, 1, 1)):
    """Triangle mesh of an occupancy volume [1,1,D,H,W] (or [D,H,W]): (verts [V,3] f32 world
    coordinates, faces [F,3] int32), or (None, None) for ``sdf is None``.  Not part of the
    reference (it renders from the volume directly); the mesh output the north star asks for.
    One host sync (the two counts); retried once with exact capacities if the guess was short."""
    if sdf is None:
        return None, None
    verts, faces, counts = ops.marching_cubes_raw(sdf, level, b_min, b_max)
    nv, nf = (int(c) for c in counts.cpu())
    if nv > verts.shape[0] or nf > faces.shape[0]:
        verts, faces, counts = ops.marching_cubes_raw(sdf, level, b_min, b_max, max_verts=nv,
                                                      max_faces=nf)
    return verts[:nv], faces[:nf]


@torch.no_grad()
def prepare_inputs(segm, mean, std, with_color=True):
    """The two "update input by removing bg" processors of RTL/main.py:352-364 as one HIP kernel:
    returns (input_netG, input_netC)."""
    return ops.prepare_inputs(segm, mean, std, with_color)


@torch.no_grad()
def visulization(render_norm, render_tex=None, render_size=256):
    """(sic) RTL/main.py:252-281: both renders scaled to 0..255, rotated by 90 degrees,
    nearest-resized to 256x256 and moved to the host as [256,256,3] numpy arrays, plus the
    foreground mask (pixels that are not pure white) of the last render present."""
    if render_norm is None and render_tex is None:
        return None, None, None
    outs, mask = [], None
    for img in (render_norm, render_tex):
        if img is None:
            outs.append(None)
            continue
        out, m = ops.visualize(img.detach(), render_size)
        outs.append(out.cpu().numpy())
        mask = m
    return outs[0], outs[1], mask.cpu().numpy().astype(bool).reshape(render_size, render_size, 1)
