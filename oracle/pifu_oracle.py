"""CPU oracle for the MonoPort reconstruction hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package (monoport_amd/) never does and fails loudly when its
HIP extension is missing.

Parity status
-------------
* ``query`` / ``sample`` (C, oracle/c/) and ``orthogonal``, ``pifu_calib``,
  ``forward_vertices``, ``colorization`` (numpy): PINNED -- checked against golden vectors
  produced by running the reference's own Python modules (oracle/gen_golden.py ->
  tests/golden/*.npz, tests/test_oracle_golden.py).
* ``seg3d_lossless`` (octree driver) and ``marching_cubes``: PARITY UNPINNED.  The reference
  delegates the octree to the un-vendored, un-pinned dependency ``implicit-seg``
  (requirements.txt:15; call sites RTL/main.py:28-29,188-195,392-394) and has no marching
  cubes at all.  Both are OUR restatements: the octree follows the published
  coarse-to-fine scheme as recalled in SURVEY.md section 5.7 and is anchored by the "lossless"
  property (thresholded octree volume == thresholded dense evaluation on smooth bodies).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libpifu_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/c/ with gcc (recipe: oracle/Makefile)."""
    # make decides about staleness (sources and the Makefile itself are prerequisites)
    subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []),
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        for suf in ("_f32", "_f64"):
            getattr(_lib, "orc_query" + suf).restype = ctypes.c_int
            getattr(_lib, "orc_sample" + suf).restype = ctypes.c_int
            getattr(_lib, "orc_orthogonal" + suf).restype = ctypes.c_int
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return _load().orc_num_threads()


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


# ------------------------------------------------------------------------------------------
# query path (C)
# ------------------------------------------------------------------------------------------
def sample(feat, uv, precision="f64"):
    """index(feat[C,H,W], uv[2,N]) -> [C,N]   (monoport/lib/modeling/geometry.py:4-16)."""
    lib = _load()
    feat = np.ascontiguousarray(feat, np.float32)
    uv = np.ascontiguousarray(uv, np.float32)
    c, h, w = feat.shape
    n = uv.shape[1]
    out = np.empty((c, n), np.float32)
    rc = getattr(lib, "orc_sample_" + precision)(
        _fptr(feat), c, h, w, _fptr(uv), ctypes.c_int64(n), _fptr(out))
    if rc != 0:
        raise RuntimeError("orc_sample failed: %d" % rc)
    return out


def query(feat, points, calib, layers, last_op, z_scale, precision="f64", threads=0):
    """MonoPortNet.query restatement (monoport/lib/modeling/MonoPortNet.py:48-91).

    feat [C,H,W] f32; points [3,N] f32 world coords; calib [>=3,4] (rows 0-2 used,
    geometry.py:27-28); layers = [(W[out,in], b[out])]; last_op 0 none / 1 sigmoid / 2 tanh.
    Returns [Cout,N] f32 (out-of-image points exactly 0, MonoPortNet.py:89).
    """
    lib = _load()
    feat = np.ascontiguousarray(feat, np.float32)
    points = np.ascontiguousarray(points, np.float32)
    assert points.shape[0] == 3
    c, h, w = feat.shape
    n = points.shape[1]
    calib12 = np.ascontiguousarray(np.asarray(calib, np.float32)[:3, :4]).reshape(12)
    ws = [np.ascontiguousarray(wl, np.float32) for wl, _ in layers]
    bs = [np.ascontiguousarray(bl, np.float32) for _, bl in layers]
    dims = [ws[0].shape[1]] + [wl.shape[0] for wl in ws]
    for l in range(1, len(ws)):
        assert ws[l].shape[1] == dims[l] + dims[0], "skip-concat width mismatch"
    nl = len(ws)
    dims_c = (ctypes.c_int * (nl + 1))(*dims)
    fpp = ctypes.POINTER(ctypes.c_float)
    w_c = (fpp * nl)(*[_fptr(a) for a in ws])
    b_c = (fpp * nl)(*[_fptr(a) for a in bs])
    out = np.empty((dims[-1], n), np.float32)
    rc = getattr(lib, "orc_query_" + precision)(
        _fptr(feat), c, h, w, _fptr(points), ctypes.c_int64(n), ctypes.c_int64(1),
        ctypes.c_int64(n), _fptr(calib12), ctypes.c_float(z_scale), nl, dims_c, w_c, b_c,
        int(last_op), _fptr(out), int(threads))
    if rc != 0:
        raise RuntimeError("orc_query failed: %d" % rc)
    return out


# ------------------------------------------------------------------------------------------
# small numpy restatements
# ------------------------------------------------------------------------------------------
def orthogonal(points, calib, precision="f32"):
    """trans + rot @ points  (monoport/lib/modeling/geometry.py:27-29); points [3,N].

    fp32: the exact op sequence of torch.baddbmm on the reference's CPU path (MKL sgemm):
    t + fma(r2, z, fma(r1, y, r0 * x)) -- bit-identical to the reference (golden test)."""
    lib = _load()
    points = np.ascontiguousarray(points, np.float32)
    assert points.shape[0] == 3
    n = points.shape[1]
    calib12 = np.ascontiguousarray(np.asarray(calib, np.float32)[:3, :4]).reshape(12)
    out = np.empty((3, n), np.float32)
    rc = getattr(lib, "orc_orthogonal_" + precision)(_fptr(points), ctypes.c_int64(n),
                                                     _fptr(calib12), _fptr(out))
    if rc != 0:
        raise RuntimeError("orc_orthogonal failed: %d" % rc)
    return out


def pifu_calib(extrinsic, intrinsic):
    """RTL/recon.py:5-25 -> [1,4,4] f32 (inputs are not mutated, :14,:17)."""
    flip = np.diag([1.0, -1.0, 1.0, 1.0])
    k = np.array(intrinsic, dtype=np.float64, copy=True)
    k[2, 2] = k[0, 0]
    k[2, 3] = 0
    e = np.array(extrinsic, copy=True)
    e[2, 3] = 0
    return np.linalg.inv(k @ e @ flip).astype(np.float32)[None]


def forward_vertices(sdf, direction="front"):
    """RTL/recon.py:27-89 on a [1,1,D,H,W] (or [D,H,W]) f32 volume.

    Returns X, Y (int64 [N]), Z (f32 [N]), norm (f32 [N,3]); all None when sdf is None (:32-33).
    For each (x, y) column the kept voxel is the first one > 0.5 along z' = R-1-z (:51-60);
    rows come out x-major (:62); neighbours sit 2 voxels back, clamped at 0 (:63-68).
    """
    if sdf is None:
        return None, None, None, None
    v = np.asarray(sdf, np.float32)
    if v.ndim == 5:
        v = v[0, 0]
    res = v.shape[2]
    if direction == "left":
        v = v.transpose(2, 1, 0)
    elif direction == "back":
        v = v[::-1]
    elif direction == "right":
        v = v[::-1].transpose(2, 1, 0)
    elif direction != "front":
        raise ValueError(direction)
    s = v[::-1].transpose(2, 1, 0)  # s[x, y, z']
    occ = s > 0.5
    hit = occ.any(axis=2)
    first = occ.argmax(axis=2)
    xs, ys = np.nonzero(hit)
    z1 = first[xs, ys]
    z2 = np.clip(z1 - 2, 0, res)
    y2 = np.clip(ys - 2, 0, res)
    x2 = np.clip(xs - 2, 0, res)
    v1 = s[xs, ys, z1]
    v2 = s[xs, ys, z2]
    v3 = s[xs, y2, z1]
    v4 = s[x2, ys, z1]
    half = np.float32(0.5)
    with np.errstate(divide="ignore", invalid="ignore"):
        zz = (z2.astype(np.float32) * (half - v1) / (v2 - v1)
              + z1.astype(np.float32) * (v2 - half) / (v2 - v1))
        zz = np.clip(zz, np.float32(0), np.float32(res)).astype(np.float32)
        nrm = np.stack([v4 - v1, v3 - v1, v2 - v1], 1).astype(np.float32)
        length = np.sqrt((nrm * nrm).sum(1, keepdims=True, dtype=np.float32))
        nrm = nrm / length
    return xs.astype(np.int64), ys.astype(np.int64), zz, nrm.astype(np.float32)


def color_matrix(b_min, b_max, res):
    """voxel -> world matrix of RTL/main.py:204-210 (scale (b_max-b_min)/res, translate b_min)."""
    m = np.eye(4, dtype=np.float32)
    length = np.asarray(b_max, np.float32) - np.asarray(b_min, np.float32)
    for i in range(3):
        m[i, i] = length[i] / np.float32(res)
    m[0:3, 3] = b_min
    return m


def colorization(X, Y, Z, res, norm=None, color_query=None, mat_color=None):
    """RTL/main.py:212-249.  ``color_query(points[3,N]) -> [3,N]`` stands for netC.query."""
    if X is None:
        return None
    canvas = np.ones((res, res, 3), np.float32)
    if norm is not None:
        canvas[X, Y, :] = np.clip((norm + np.float32(1)) / np.float32(2), 0, 1)
        return canvas
    verts = np.stack([X.astype(np.float32), Y.astype(np.float32),
                      np.float32(res) - Z.astype(np.float32)], 0)
    samples = orthogonal(verts, mat_color)
    preds = color_query(samples)
    canvas[X, Y, :] = (preds * np.float32(0.5) + np.float32(0.5)).T
    return canvas


# ------------------------------------------------------------------------------------------
# octree driver (OUR restatement of implicit_seg.Seg3dLossless(faster=True); parity unpinned)
# ------------------------------------------------------------------------------------------
def lattice_points(idx_zyx, stride, res_final, b_min, b_max):
    """Index (z,y,x) at a level of spacing ``stride`` -> world [3,N] f32.

    p = ((c / R) + (1/R)/2) * (b_max - b_min) + b_min with c = index * stride in final-resolution
    index space, every step rounded to f32 (align_corners=False convention).
    """
    r = np.float32(res_final)
    half_step = np.float32(np.float32(1.0) / r) / np.float32(2)
    b_min = np.asarray(b_min, np.float32).reshape(3)
    b_max = np.asarray(b_max, np.float32).reshape(3)
    out = np.empty((3, idx_zyx.shape[0]), np.float32)
    for axis, col in ((0, 2), (1, 1), (2, 0)):  # x <- idx[:,2], y <- idx[:,1], z <- idx[:,0]
        c = (idx_zyx[:, col].astype(np.int64) * stride).astype(np.float32)
        u = (c / r + half_step).astype(np.float32)
        out[axis] = (u * (b_max[axis] - b_min[axis]) + b_min[axis]).astype(np.float32)
    return out


def upsample2x(a):
    """Trilinear upsample r -> 2r-1 with align_corners=True: even nodes copy, odd nodes average."""
    a = np.asarray(a, np.float32)
    half = np.float32(0.5)
    for axis in range(3):
        n = a.shape[axis]
        shape = list(a.shape)
        shape[axis] = 2 * n - 1
        out = np.empty(shape, np.float32)
        ev = [slice(None)] * 3
        od = [slice(None)] * 3
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        ev[axis] = slice(0, None, 2)
        od[axis] = slice(1, None, 2)
        lo[axis] = slice(0, n - 1)
        hi[axis] = slice(1, n)
        out[tuple(ev)] = a
        out[tuple(od)] = half * a[tuple(lo)] + half * a[tuple(hi)]
        a = out
    return a


def dilate_box(mask, k):
    """(all-ones k^3 conv3d, zero padding k//2) > 0 on a bool volume."""
    r = k // 2
    m = mask
    for axis in range(3):
        acc = m.copy()
        n = m.shape[axis]
        for s in range(1, r + 1):
            a = [slice(None)] * 3
            b = [slice(None)] * 3
            a[axis] = slice(s, n)
            b[axis] = slice(0, n - s)
            acc[tuple(a)] |= m[tuple(b)]
            acc[tuple(b)] |= m[tuple(a)]
        m = acc
    return m


def dilation_for_level(level):
    """Box size per level in 'faster' mode: 9^3 at level 1, 7^3 at level 2, 3^3 after."""
    return {1: 9, 2: 7}.get(level, 3)


FINAL_LEVELS = ("dilate3", "upstream", "interpolate")


def seg3d_lossless(query_func, b_min, b_max, resolutions, balance_value=0.5, stats=None,
                   evaluated_out=None, faster=True, rounds=None, final_level="dilate3"):
    """Coarse-to-fine occupancy volume [R,R,R] (z,y,x) f32, or None if level 0 is empty.

    ``query_func(points[3,N] f32) -> [N] f32``.  Requires resolutions[i+1] == 2*resolutions[i]-1.
    ``evaluated_out`` (bool [R,R,R] of the final resolution) receives the set of queried nodes.
    ``faster=True``: dilation boxes 9/7/3 by level, no re-examination (the mode RTL/main.py:194
    selects).  ``faster=False``: 3^3 boxes at every level and the conflict loop -- a node whose
    exact value and interpolated value lie on different sides of the threshold gets its 3x3x3
    neighbourhood (at this level's spacing) evaluated as well, repeated until no new conflict;
    ``stats`` counts those points with their level, ``rounds`` receives the rounds per level.
    ``final_level`` (faster=True): the selection rule of the LAST level -- "dilate3": like every level
    >= 3 (the lossless schedule); "upstream": only nodes whose upsampled inside-mask is exactly 0.5, no
    dilation (``is_boundary = valid == 0.5``, the rule recalled from the upstream package's faster
    mode); "interpolate": nothing is evaluated there (its "last step no examine").  Unpinned by the
    reference either way.
    """
    if final_level not in FINAL_LEVELS or (final_level != "dilate3" and not faster):
        raise ValueError("final_level %r" % (final_level,))
    res = [int(r) for r in resolutions]
    for a, b in zip(res[:-1], res[1:]):
        if b != 2 * a - 1:
            raise ValueError("resolutions must follow r -> 2r-1")
    rf = res[-1]
    r0 = res[0]
    bv = np.float32(balance_value)
    stride = (rf - 1) // (r0 - 1)
    idx = np.stack(np.meshgrid(np.arange(r0), np.arange(r0), np.arange(r0), indexing="ij"),
                   -1).reshape(-1, 3)
    occ = np.asarray(query_func(lattice_points(idx, stride, rf, b_min, b_max)),
                     np.float32).reshape(r0, r0, r0)
    if stats is not None:
        stats.append(idx.shape[0])
    if rounds is not None:
        rounds.append(0)
    if not (occ > bv).any():
        return None
    evaluated = np.ones((r0, r0, r0), bool)
    for level in range(1, len(res)):
        r = res[level]
        stride = (rf - 1) // (r - 1)
        valid = upsample2x((occ > bv).astype(np.float32))
        occ = upsample2x(occ)
        last = faster and level == len(res) - 1
        if last and final_level == "upstream":
            sel = valid == np.float32(0.5)
        elif last and final_level == "interpolate":
            sel = np.zeros((r, r, r), bool)
        else:
            boundary = (valid > 0) & (valid < 1)
            sel = dilate_box(boundary, dilation_for_level(level) if faster else 3)
        ev = np.zeros((r, r, r), bool)
        ev[::2, ::2, ::2] = evaluated
        sel &= ~ev
        idx = np.argwhere(sel)
        n_level, n_rounds = idx.shape[0], 0
        evaluated = ev | sel
        while idx.shape[0]:
            vals = np.asarray(query_func(lattice_points(idx, stride, rf, b_min, b_max)),
                              np.float32)
            interp = occ[idx[:, 0], idx[:, 1], idx[:, 2]]
            occ[idx[:, 0], idx[:, 1], idx[:, 2]] = vals
            if faster:
                break
            conflict = ((interp - bv) * (vals - bv)) < 0
            if not conflict.any():
                break
            grow = np.zeros((r, r, r), bool)
            c = idx[conflict]
            grow[c[:, 0], c[:, 1], c[:, 2]] = True
            grow = dilate_box(grow, 3) & ~evaluated
            idx = np.argwhere(grow)
            evaluated |= grow
            n_level += idx.shape[0]
            n_rounds += 1 if idx.shape[0] else 0
        if stats is not None:
            stats.append(n_level)
        if rounds is not None:
            rounds.append(n_rounds)
    if evaluated_out is not None:
        evaluated_out[...] = evaluated
    return occ


def dense_volume(query_func, b_min, b_max, res, res_final=None):
    """Dense evaluation of every lattice node at resolution ``res`` ([z,y,x] f32)."""
    rf = res if res_final is None else res_final
    stride = (rf - 1) // (res - 1) if res > 1 else 1
    idx = np.stack(np.meshgrid(np.arange(res), np.arange(res), np.arange(res), indexing="ij"),
                   -1).reshape(-1, 3)
    return np.asarray(query_func(lattice_points(idx, stride, rf, b_min, b_max)),
                      np.float32).reshape(res, res, res)


# ------------------------------------------------------------------------------------------
# marching cubes (OUR variant -- the reference has none; table from tools/gen_mc_tables.py)
# ------------------------------------------------------------------------------------------
_mc = None


def _mc_tables():
    global _mc
    if _mc is None:
        t = np.load(os.path.join(_HERE, "mc_tables.npz"))
        _mc = {k: t[k] for k in t.files}
    return _mc


def marching_cubes(vol, level=0.5, b_min=(-1, -1, -1), b_max=(1, 1, 1)):
    """Mesh of volume [R,R,R] (z,y,x): verts [V,3] f32 world coords, faces [F,3] int32.

    One welded vertex per lattice edge whose endpoints straddle ``level`` (inside = value >
    level), ordered by edge id = ((z*R+y)*R+x)*3 + axis (axis 0 = x); placed at the linear crossing
    t = (level - va)/(vb - va) and mapped to world space with the octree lattice convention
    ((p/R) + (1/R)/2) * (b_max-b_min) + b_min.  Triangles: cells in (z,y,x) order, table order
    within a cell, counter-clockwise seen from outside.
    """
    t = _mc_tables()
    vol = np.asarray(vol, np.float32)
    r = vol.shape[0]
    lvl = np.float32(level)
    inside = vol > lvl
    vidx = -np.ones((r, r, r, 3), np.int64)
    cross = np.zeros((r, r, r, 3), bool)
    cross[:, :, :-1, 0] = inside[:, :, :-1] != inside[:, :, 1:]
    cross[:, :-1, :, 1] = inside[:, :-1, :] != inside[:, 1:, :]
    cross[:-1, :, :, 2] = inside[:-1, :, :] != inside[1:, :, :]
    flat = cross.reshape(-1)
    vidx.reshape(-1)[flat] = np.arange(int(flat.sum()))
    zz, yy, xx, aa = np.nonzero(cross)  # ascending edge id
    va = vol[zz, yy, xx]
    vb = vol[zz + (aa == 2), yy + (aa == 1), xx + (aa == 0)]
    tt = ((lvl - va) / (vb - va)).astype(np.float32)
    pos = np.stack([xx, yy, zz], 1).astype(np.float32)
    pos[np.arange(pos.shape[0]), aa] = pos[np.arange(pos.shape[0]), aa] + tt
    rf = np.float32(r)
    half_step = np.float32(np.float32(1.0) / rf) / np.float32(2)
    bmin = np.asarray(b_min, np.float32).reshape(3)
    blen = np.asarray(b_max, np.float32).reshape(3) - bmin
    verts = ((pos / rf + half_step).astype(np.float32) * blen + bmin).astype(np.float32)

    case = np.zeros((r - 1, r - 1, r - 1), np.int32)
    for i in range(8):
        dx, dy, dz = i & 1, (i >> 1) & 1, (i >> 2) & 1
        case |= inside[dz:r - 1 + dz, dy:r - 1 + dy, dx:r - 1 + dx].astype(np.int32) << i
    cz, cy, cx = np.nonzero(t["count"][case] > 0)
    cc = case[cz, cy, cx]
    faces = []
    for k in range(t["tri"].shape[1]):
        sel = t["count"][cc] > k
        if not sel.any():
            break
        tri = t["tri"][cc[sel], k]  # [n,3] edge ids
        own = t["owner"][tri]       # [n,3,4]: dx,dy,dz,axis
        f = vidx[cz[sel, None] + own[..., 2], cy[sel, None] + own[..., 1],
                 cx[sel, None] + own[..., 0], own[..., 3]]
        order = np.nonzero(sel)[0] * 8 + k  # cell-major, then table order
        faces.append((order, f))
    if faces:
        order = np.concatenate([o for o, _ in faces])
        f = np.concatenate([f for _, f in faces])
        f = f[np.argsort(order, kind="stable")]
    else:
        f = np.zeros((0, 3), np.int64)
    assert (f >= 0).all()
    return verts, f.astype(np.int32)
