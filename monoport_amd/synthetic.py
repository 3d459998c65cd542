"""Seeded synthetic fixtures for the PIFu query hot path (numpy only, no torch, no GPU).

The reference ships no weights (scripts/download_model.sh needs the network) and no test
vectors, so parity tests, the golden-vector generator and bench.py all
draw their inputs from here:

* ``rand_mlp``  -- F-rand: every weight of the SurfaceClassifier MLP exercised
  (shapes from monoport/lib/modeling/heads/SurfaceClassifier.py:74-87).
* ``body_mlp`` + ``body_feature_planes`` -- F-body: analytic weights that turn two feature
  channels (front / back depth maps of a capsule figure) into a smooth closed occupancy
  field, so the octree sees a realistic surface.
* ``scene_camera`` -- the orbit camera of RTL/scene.py:45-50,122-137 restated numerically.

All generators use ``numpy.random.RandomState`` so values are identical on every machine.
"""
import math

import numpy as np

# monoport/lib/modeling/heads/SurfaceClassifier.py:76 / :84
MLP_DIMS = {
    "G": [257, 1024, 512, 256, 128, 1],
    "C": [513, 1024, 512, 256, 128, 3],
}
# last_op codes shared with the C-ABI (include/monoport_hip.h)
LAST_OP = {"G": 1, "C": 2}  # 1 = sigmoid (SurfaceClassifier.py:77), 2 = tanh (:85)
# monoport/lib/modeling/normalizers/DepthNormalizer.py:40
Z_SCALE = 512 // 2 / 200.0


def layer_shapes(kind):
    """[(out, in)] per layer with the skip-concat widths (SurfaceClassifier.py:26-31)."""
    d = MLP_DIMS[kind]
    return [(d[l + 1], d[l] + (d[0] if l > 0 else 0)) for l in range(len(d) - 1)]


def rand_mlp(kind, seed, gain=1.0):
    """F-rand: list of (W[out,in] f32, b[out] f32), uniform(+-gain/sqrt(fan_in))."""
    rs = np.random.RandomState(seed)
    layers = []
    for out_c, in_c in layer_shapes(kind):
        bound = gain / math.sqrt(in_c)
        w = rs.uniform(-bound, bound, size=(out_c, in_c)).astype(np.float32)
        b = rs.uniform(-bound, bound, size=(out_c,)).astype(np.float32)
        layers.append((w, b))
    return layers


def rand_feat(c, h, w, seed, scale=1.0):
    """Seeded feature map [C,H,W] f32 (the layout MonoPortNet.filter emits, HGFilters.py:196)."""
    rs = np.random.RandomState(seed)
    return (rs.standard_normal((c, h, w)) * scale).astype(np.float32)


def rand_points(n, seed, extent=1.2):
    """[3,N] f32 world points in [-extent, extent]^3 (extent > 1 exercises the out-of-image mask)."""
    rs = np.random.RandomState(seed)
    return rs.uniform(-extent, extent, size=(3, n)).astype(np.float32)


# ------------------------------------------------------------------------------------------
# F-body: a capsule figure described by front/back depth maps in image space
# ------------------------------------------------------------------------------------------
_PARTS = [
    # (cx, cy, cz, rx, ry, rz) ellipsoids in image space (x right, y down, z toward camera)
    (0.00, -0.62, 0.00, 0.14, 0.17, 0.15),  # head
    (0.00, -0.15, 0.00, 0.26, 0.36, 0.17),  # torso
    (-0.36, -0.18, 0.02, 0.09, 0.34, 0.09),  # left arm
    (0.36, -0.18, 0.02, 0.09, 0.34, 0.09),  # right arm
    (-0.13, 0.50, 0.00, 0.11, 0.40, 0.12),  # left leg
    (0.13, 0.50, 0.00, 0.11, 0.40, 0.12),  # right leg
]
# further figures for the octree tests: thin limbs (radius ~2 voxels at 257^3), two separate bodies
_PARTS_THIN = [
    (0.00, -0.55, 0.00, 0.10, 0.12, 0.10),   # head
    (0.00, -0.10, 0.00, 0.05, 0.40, 0.05),   # spine
    (-0.30, -0.20, 0.03, 0.018, 0.36, 0.018),  # very thin arms / legs
    (0.30, -0.20, 0.03, 0.018, 0.36, 0.018),
    (-0.10, 0.55, 0.00, 0.02, 0.35, 0.02),
    (0.10, 0.55, 0.00, 0.02, 0.35, 0.02),
    (0.00, -0.30, 0.00, 0.34, 0.015, 0.015),  # thin horizontal bar (shoulders)
]
_PARTS_TWO = [
    (-0.45, -0.10, -0.30, 0.22, 0.55, 0.20),  # two disconnected bodies at different depths
    (0.48, 0.15, 0.35, 0.18, 0.42, 0.16),
    (0.48, -0.45, 0.35, 0.10, 0.12, 0.10),
]
FIGURES = {"figure": _PARTS, "thin": _PARTS_THIN, "two": _PARTS_TWO}
_EMPTY_FRONT = -4.0  # depth planes where no part covers the pixel: z_f < z_b -> always outside
_EMPTY_BACK = 4.0


def body_depth_maps(h=128, w=128, figure="figure"):
    """Front (max z) and back (min z) depth maps [H,W] f32 of a capsule figure (``FIGURES``).

    Pixel (i, j) sits at x = -1 + 2j/(W-1), y = -1 + 2i/(H-1): the align_corners=True grid
    of monoport/lib/modeling/geometry.py:15.
    """
    ys = np.linspace(-1.0, 1.0, h)[:, None]
    xs = np.linspace(-1.0, 1.0, w)[None, :]
    zf = np.full((h, w), _EMPTY_FRONT, dtype=np.float64)
    zb = np.full((h, w), _EMPTY_BACK, dtype=np.float64)
    for cx, cy, cz, rx, ry, rz in FIGURES[figure]:
        q = 1.0 - ((xs - cx) / rx) ** 2 - ((ys - cy) / ry) ** 2
        inside = q > 0
        dz = rz * np.sqrt(np.where(inside, q, 0.0))
        zf = np.where(inside, np.maximum(zf, cz + dz), zf)
        zb = np.where(inside, np.minimum(zb, cz - dz), zb)
    return zf.astype(np.float32), zb.astype(np.float32)


def body_feature_planes(h=128, w=128, figure="figure"):
    """The two feature channels F-body reads: channel 0 = z_front, channel 1 = z_back."""
    zf, zb = body_depth_maps(h, w, figure)
    return np.stack([zf, zb], 0)


def body_mlp(kind="G", k=40.0, c=2.0, noise=0.0, seed=0):
    """F-body analytic weights.

    layer 0:  h0 = k (z - z_front),  h1 = k (z_back - z)      (z = z_feat / Z_SCALE)
    layers 1-3 pass hidden units 0/1 through (leaky-ReLU keeps positives, SurfaceClassifier.py:58)
    last:     y = c - h0 - h1  -> sigmoid: > 0.5 strictly inside, < 0.5 outside.
    ``noise`` adds seeded uniform(+-noise/sqrt(fan_in)) to every weight so all MFMA operands
    are exercised without moving the surface much.
    """
    shapes = layer_shapes(kind)
    rs = np.random.RandomState(seed)
    layers = []
    n_feat = MLP_DIMS[kind][0]  # C + 1; z_feat is the LAST input channel (MonoPortNet.py:83)
    for l, (out_c, in_c) in enumerate(shapes):
        if noise > 0:
            bound = noise / math.sqrt(in_c)
            w = rs.uniform(-bound, bound, size=(out_c, in_c)).astype(np.float32)
            b = rs.uniform(-bound, bound, size=(out_c,)).astype(np.float32)
        else:
            w = np.zeros((out_c, in_c), np.float32)
            b = np.zeros((out_c,), np.float32)
        if l == 0:
            w[0, :] = 0
            w[1, :] = 0
            w[0, 0] = -k  # -k z_front
            w[0, n_feat - 1] = k / Z_SCALE  # +k z
            w[1, 1] = k  # +k z_back
            w[1, n_feat - 1] = -k / Z_SCALE  # -k z
            b[0] = b[1] = 0
        elif l < len(shapes) - 1:
            w[0, :] = 0
            w[1, :] = 0
            w[0, 0] = 1.0
            w[1, 1] = 1.0
            b[0] = b[1] = 0
        else:
            w[0, 0] -= 1.0
            w[0, 1] -= 1.0
            b[0] += c
        layers.append((w, b))
    return layers


def readout_body_mlp(readout, r0, thick, k=40.0, c=2.0, noise=0.0, seed=0):
    """F-body on ENCODER features: the slab |z| < t(x, y) whose half thickness is a linear readout of the
    256 feature channels, t = thick * (readout . feat - r0) -- positive where the readout exceeds r0
    (inside the silhouette of the input image for a readout fitted to the encoder's output, see
    oracle/gen_golden.py: gen_pipeline257_color), negative elsewhere (always outside).

    layer 0:  h0 = k (z - t),  h1 = k (-t - z)          (z = z_feat / Z_SCALE)
    layers 1-3 / last layer / ``noise`` as in ``body_mlp``.  Every feature channel moves the surface:
    the encoder is in the loop of every octree decision."""
    layers = body_mlp("G", k=k, c=c, noise=noise, seed=seed)
    w0, b0 = layers[0]
    r = np.asarray(readout, np.float64).reshape(256)
    for row, sign in ((0, 1.0), (1, -1.0)):
        w0[row, :] = 0
        w0[row, :256] = (-k * thick * r).astype(np.float32)
        w0[row, 256] = np.float32(sign * k / Z_SCALE)
        b0[row] = np.float32(k * thick * r0)
    return layers


def body_feat(c=256, h=128, w=128, seed=0, scale=1.0, figure="figure"):
    """Seeded feature map whose channels 0/1 carry the body depth planes."""
    f = rand_feat(c, h, w, seed, scale)
    f[0:2] = body_feature_planes(h, w, figure)
    return f


def synthetic_image(seed, size=512):
    """[3,size,size] f32 masked silhouette in [-1,1] with the background zeroed,
    the shape netG.filter sees after RTL/main.py:353-357."""
    rs = np.random.RandomState(seed)
    zf, _ = body_depth_maps(size, size)
    mask = (zf > _EMPTY_FRONT + 1).astype(np.float32)
    img = rs.uniform(-1.0, 1.0, size=(3, size, size)).astype(np.float32)
    return img * mask[None]


# ------------------------------------------------------------------------------------------
# camera (RTL/scene.py) -- the matrices pifu_calib consumes
# ------------------------------------------------------------------------------------------
def _rot_xyz(rx, ry, rz):
    """Rz @ Ry @ Rx, the composition of RTL/scene.py:62-93."""
    sx, cx = math.sin(rx), math.cos(rx)
    sy, cy = math.sin(ry), math.cos(ry)
    sz, cz = math.sin(rz), math.cos(rz)
    mx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float64)
    my = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    mz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float64)
    return mz @ my @ mx


def scene_camera(step=0, yaw_deg=20.0):
    """(extrinsic[4,4] f32, intrinsic[4,4] f64) of the orbit camera at ``step``.

    RTL/scene.py:108-113 (extrinsic template, z = -2), :124-136 (yaw about x by 20 deg, pitch
    about y by ``step`` deg), :45-50 + BaseCamera.py:47-54 (ortho projection, near 0, far 10,
    magnification 2).
    """
    ext = np.eye(4, dtype=np.float32)
    ext[2, 3] = -2.0
    rot = _rot_xyz(math.radians(yaw_deg), 0, 0) @ _rot_xyz(0, math.radians(step), 0)
    ext[0:3, 0:3] = rot
    near, far, mag = 0.0, 10.0, 2.0
    intr = np.eye(4, dtype=np.float64)
    intr[0, 0] = 2 / mag
    intr[1, 1] = 2 / mag
    intr[2, 2] = -2 / (far - near)
    intr[2, 3] = -(far + near) / (far - near)
    return ext, intr


def blob_volume(res, seed, n_blobs=6, sharp=6.0):
    """Smooth seeded occupancy-like volume [res,res,res] (z,y,x) f32 in (0,1) with a one-voxel
    empty shell, so forward_vertices never divides by zero at z' = 0."""
    rs = np.random.RandomState(seed)
    g = np.linspace(-1.0, 1.0, res)
    zz, yy, xx = np.meshgrid(g, g, g, indexing="ij")
    field = np.full((res, res, res), -1.0)
    for _ in range(n_blobs):
        c = rs.uniform(-0.45, 0.45, size=3)
        r = rs.uniform(0.15, 0.35)
        d = np.sqrt((xx - c[0]) ** 2 + (yy - c[1]) ** 2 + (zz - c[2]) ** 2)
        field = np.maximum(field, (r - d) / r)
    vol = 1.0 / (1.0 + np.exp(-sharp * field))
    vol = vol + rs.uniform(-1e-3, 1e-3, size=vol.shape)
    edge = np.zeros_like(vol, bool)
    edge[[0, -1], :, :] = edge[:, [0, -1], :] = edge[:, :, [0, -1]] = True
    vol[edge] = 0.01
    return vol.astype(np.float32)


def seeded_state_dict(shapes, seed):
    """Deterministic weights for an encoder: ``shapes`` = {state_dict key: shape}.

    Each tensor is drawn from a RandomState seeded by (seed, crc32(key)), so the result does not
    depend on key order.  Keys that alias one parameter in the reference (``downsample.0.*`` is
    ``bn4.*``, HGFilters.py:29-35) get identical values.  Conv weights ~ N(0, 2/fan_in), norm
    scales ~ 1 + 0.1 N, biases ~ 0.1 N.
    """
    import zlib
    out = {}
    for key, shape in shapes.items():
        canon = key.replace("downsample.0.", "bn4.")
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(canon.encode())) % (2 ** 32))
        shape = tuple(shape)
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = rs.standard_normal(shape) * math.sqrt(2.0 / fan_in)
        elif key.endswith("weight"):
            v = 1.0 + 0.1 * rs.standard_normal(shape)
        else:
            v = 0.1 * rs.standard_normal(shape)
        out[key] = v.astype(np.float32)
    return out


def obj_mesh_inputs():
    """A seeded mesh for the on-disk format check: 500 vertices with coordinates that exercise the
    %.4f rounding (ties, negative zero, values below 5e-5, large magnitudes), 900 faces, colours in
    [0, 1].  Regenerated by the tests from the same seed."""
    rng = np.random.RandomState(4242)
    v = (rng.standard_normal((500, 3)) * np.array([1.0, 100.0, 1e-3])).astype(np.float32)
    v[:8, 0] = np.array([0.00005, -0.00005, 0.12345, -0.12345, 0.99995, -0.0, 1e-9, 12345.67895], np.float32)
    f = rng.randint(0, 500, size=(900, 3)).astype(np.int32)
    c = rng.rand(500, 3).astype(np.float32)
    return v, f, c
