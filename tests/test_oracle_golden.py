"""Pin the CPU oracle against golden vectors produced by the reference's own modules
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from monoport_amd import synthetic as syn

QUERY_CASES = {
    # >= 40 k points each, > 98 % of them inside the image (the rest exercise the exact-zero mask)
    "query_G_rand": ("G", ("rand", 11, 2.0), ("rand", 256, 21), (49152, 31, 0.8)),
    "query_C_rand": ("C", ("rand", 12, 2.0), ("rand", 512, 22), (40960, 32, 0.8)),
    "query_G_body": ("G", ("body", 13, 0.05), ("body", 256, 23), (40960, 33, 0.7)),
}


def query_inputs(name):
    kind, mlp, feat, pts = QUERY_CASES[name]
    layers = (syn.rand_mlp(kind, mlp[1], mlp[2]) if mlp[0] == "rand"
              else syn.body_mlp(kind, noise=mlp[2], seed=mlp[1]))
    f = (syn.rand_feat(feat[1], 128, 128, feat[2]) if feat[0] == "rand"
         else syn.body_feat(feat[1], 128, 128, feat[2]))
    p = syn.rand_points(*pts)
    return kind, layers, f, p


@pytest.mark.parametrize("name", sorted(QUERY_CASES))
@pytest.mark.parametrize("precision,tol", [("f64", 5e-5), ("f32", 5e-6)])
def test_query_matches_reference(oracle, name, precision, tol):
    # The fp32 oracle follows the reference's CPU op order (MKL baddbmm and torch's grid_sample
    # FMA chains): projection and sampled features are bit-identical, what is left (<= 2e-6) is
    # the summation order inside the MLP GEMMs.  The fp64 tolerance covers the fp32 noise of the
    # reference itself: 3.1e-5 on the netC fixture (K=1537, gain 2); the north-star bar is 1e-4.
    g = load_golden(name)
    kind, layers, f, p = query_inputs(name)
    out = oracle.query(f, p, g["calib"][0], layers, syn.LAST_OP[kind], syn.Z_SCALE,
                       precision=precision)
    ref = g["out"]
    assert out.shape == ref.shape
    # out-of-image points are exactly zero in the reference (MonoPortNet.py:89)
    xyz = oracle.orthogonal(p, g["calib"][0])
    margin = np.minimum(1 - np.abs(xyz[0]), 1 - np.abs(xyz[1]))
    outside = margin < -1e-6
    assert outside.sum() > 100 or name == "query_G_body"
    assert (~outside).sum() >= 32768
    assert (out[:, outside] == 0).all() and (ref[:, outside] == 0).all()
    if name == "query_G_body" and precision == "f64":
        # at the silhouette the depth planes jump by ~4 units per pixel and the head multiplies
        # by k = 40: the reference's own fp32 rounding of ix = (x+1)/2*(W-1) (~8e-6 px) is worth
        # up to ~3e-4 there.  The fp32 oracle follows the reference's op order and stays at 5e-5.
        tol = 5e-4
    assert np.abs(out - ref).max() <= tol


@pytest.mark.parametrize("name", sorted(QUERY_CASES))
def test_torch_ops_query_matches_reference(name):
    """oracle/torch_ops.py restates the reference's OPERATOR SEQUENCE (baddbmm, grid_sample, Conv1d chain) so that
    bench.py can time "the reference CPU recon path" on a box without /root/reference.  Same operators, same order:
    in the container that generated the goldens the outputs are the reference's bits; elsewhere (other core counts:
    oneDNN / MKL may block differently) within fp32 summation noise."""
    from oracle import torch_ops
    g = load_golden(name)
    kind, layers, f, p = query_inputs(name)
    out = torch_ops.query(f, p, g["calib"][0], layers, syn.LAST_OP[kind], syn.Z_SCALE)
    same = np.array_equal(out, g["out"])
    err = float(np.abs(out - g["out"]).max())
    print("%s: torch-operator restatement vs the reference: bit-identical %s, max|d| %.3g" % (name, same, err))
    assert out.shape == g["out"].shape and err <= 2e-6
    assert np.array_equal(out == 0, g["out"] == 0)


def test_torch_ops_drive_the_octree_like_the_reference(oracle):
    """... and driving the 17..257 schedule with it reproduces the reference-driven fixture node for node: same
    queried set, same per-level counts, values within fp32 summation noise (bit-identical where the goldens were made)."""
    from oracle import torch_ops
    name = "pipeline257"
    g, queried_ref = pipeline257_golden(name)
    layers, f, step = pipeline257_inputs(name)
    calib = oracle.pifu_calib(*syn.scene_camera(step))
    stats, queried = [], np.zeros_like(queried_ref)
    vol = oracle.seg3d_lossless(lambda p: torch_ops.query(f, p, calib[0], layers, 1, syn.Z_SCALE)[0],
                                [-1, -1, -1], [1, 1, 1], PIPE257_RES, stats=stats, evaluated_out=queried)
    assert stats == list(g["stats"]) and np.array_equal(queried, queried_ref)
    err = float(np.abs(vol[queried] - g["values"]).max())
    print("octree through the torch-operator restatement: max|value - reference| %.3g over %d nodes, bit-identical %s"
          % (err, int(queried.sum()), np.array_equal(vol[queried], g["values"])))
    assert err <= 2e-6


def test_index_matches_reference(oracle):
    g = load_golden("index")
    f = syn.rand_feat(256, 128, 128, 41)
    out = oracle.sample(f, g["uv"], precision="f64")
    # the fp64 result differs from the fp32 reference by the rounding of ix=((x+1)/2)*(W-1): ~3e-5 on
    # white-noise features; the fp32 oracle follows the reference's op order: identical bits
    assert np.abs(out - g["out"]).max() <= 6e-5
    out32 = oracle.sample(f, g["uv"], precision="f32")
    assert np.array_equal(out32, g["out"])


def test_orthogonal_matches_reference(oracle):
    g = load_golden("orthogonal")
    p = syn.rand_points(1000, 43, 1.0)
    assert np.array_equal(oracle.orthogonal(p, g["calib"][0]), g["out"])  # bit for bit
    assert np.abs(oracle.orthogonal(p, g["calib"][0], precision="f64") - g["out"]).max() <= 1e-6


def test_pifu_calib_matches_reference(oracle):
    g = load_golden("pifu_calib")
    for step, ref in zip(g["steps"], g["calib"]):
        ext, intr = syn.scene_camera(int(step))
        e0, i0 = ext.copy(), intr.copy()
        out = oracle.pifu_calib(ext, intr)
        assert out.shape == (1, 4, 4) and out.dtype == np.float32
        assert np.array_equal(out[0], ref)
        assert np.array_equal(e0, ext) and np.array_equal(i0, intr)  # recon.py:14,17 copies


@pytest.mark.parametrize("res,seed", [(33, 51), (65, 52)])
@pytest.mark.parametrize("direction", ["front", "back", "left", "right"])
def test_forward_vertices_matches_reference(oracle, res, seed, direction):
    g = load_golden("forward_vertices")
    vol = syn.blob_volume(res, seed)
    x, y, z, n = oracle.forward_vertices(vol[None, None], direction)
    key = "r%d_%s_" % (res, direction)
    assert x.dtype == np.int64 and y.dtype == np.int64
    assert np.array_equal(x, g[key + "X"]) and np.array_equal(y, g[key + "Y"])
    assert np.abs(z - g[key + "Z"]).max() <= 1e-4  # Z is in voxel units (0..res)
    assert np.abs(n - g[key + "norm"]).max() <= 1e-5


def test_forward_vertices_none(oracle):
    assert oracle.forward_vertices(None) == (None, None, None, None)


def test_colorization_matches_reference(oracle):
    g = load_golden("colorization")
    res = 33
    vol = syn.blob_volume(res, 63)
    x, y, z, n = oracle.forward_vertices(vol, "front")
    img_n = oracle.colorization(x, y, z, res, norm=n)
    assert np.abs(img_n - g["norm_image"]).max() <= 1e-5
    layers = syn.rand_mlp("C", 61, 2.0)
    f = syn.rand_feat(512, 128, 128, 62)
    mat = oracle.color_matrix([-1, -1, -1], [1, 1, 1], res)

    def color_query(pts):
        return oracle.query(f, pts, g["calib"][0], layers, syn.LAST_OP["C"], syn.Z_SCALE)

    img_t = oracle.colorization(x, y, z, res, color_query=color_query, mat_color=mat)
    assert np.abs(img_t - g["tex_image"]).max() <= 2e-5
    assert oracle.colorization(None, None, None, res) is None


# ---- BASELINE-size fixtures (oracle/gen_golden.py: gen_dense64, gen_pipeline257) ------------------
def dense_lattice(res):
    """p = ((i + 0.5) / res) * 2 - 1, [z,y,x] order -> [3, res^3]."""
    g = ((np.arange(res, dtype=np.float32) + np.float32(0.5)) / np.float32(res)) * np.float32(2) - np.float32(1)
    zz, yy, xx = np.meshgrid(g, g, g, indexing="ij")
    return np.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], 0).astype(np.float32)


DENSE64_CASES = {
    "out_rand": (("rand", 91, 2.0), ("rand", 92)),
    "out_body": (("body", 93, 0.05), ("body", 94)),
}


def dense64_inputs(case):
    mlp, feat = DENSE64_CASES[case]
    layers = (syn.rand_mlp("G", mlp[1], mlp[2]) if mlp[0] == "rand"
              else syn.body_mlp("G", noise=mlp[2], seed=mlp[1]))
    f = (syn.rand_feat(256, 128, 128, feat[1]) if feat[0] == "rand"
         else syn.body_feat(256, 128, 128, feat[1]))
    return layers, f


@pytest.mark.parametrize("case", sorted(DENSE64_CASES))
def test_dense64_matches_reference(oracle, case):
    """BASELINE configs[0]: all 262,144 nodes of the dense 64^3 grid through the reference's
    netG.query on the CPU vs the C oracle."""
    g = load_golden("dense64")
    layers, f = dense64_inputs(case)
    p = dense_lattice(64)
    out = oracle.query(f, p, g["calib"][0], layers, syn.LAST_OP["G"], syn.Z_SCALE, precision="f32")[0]
    ref = g[case]
    assert out.shape == ref.shape == (64 ** 3,)
    assert np.array_equal(out == 0, ref == 0) or case == "out_body"  # the in-image mask
    assert np.abs(out - ref).max() <= 5e-6


PIPE257_RES = [17, 33, 65, 129, 257]
# name -> (body_mlp arguments, body_feat seed, camera step): oracle/gen_golden.py PIPE257_SCENES.
# "pipeline257" is round 2's scene (camera picked for a 6e-6 margin to the threshold: identical
# decisions under any fp32-class evaluation); "_b" and "_soft" were NOT picked -- each holds queried
# values within one ulp of 0.5 -- and "_soft" is an unsaturated field in which every head weight and
# feature channel matters (72 % of its queried values lie in (0.01, 0.99)).
PIPE257_SCENES = {
    "pipeline257": (dict(k=40.0, c=2.0, noise=0.05, seed=95), 96, 170),
    "pipeline257_b": (dict(k=40.0, c=2.0, noise=0.05, seed=195), 196, 40),
    "pipeline257_soft": (dict(k=6.0, c=2.0, noise=1.0, seed=295), 296, 250),
}
PIPE257 = dict(mlp=("body", 95, 0.05), feat=96, step=170, res=PIPE257_RES)  # round-2 name of the first scene
# BASELINE configs[4] size (round 6; oracle/gen_golden.py PIPE513_SCENES): the body / camera of the 513^3 GPU
# tests, 17..513 through the reference's netG.query -- 1,152,942 reference-evaluated nodes, 854,388 of them
# on the level-5 lattice (10-bit packed coordinates).
PIPE513_RES = [17, 33, 65, 129, 257, 513]
PIPE513_SCENES = {
    "pipeline513": (dict(k=40.0, c=2.0, noise=0.05, seed=1), 2, 30),
    # noise 2.0: the seeded weights of every layer are 40x larger -- f16 rounding of the weights of layers 0-3
    # moves the occupancy by up to ~1e-4 (1e-7 in the scene above, whose large weights are exact in f16)
    "pipeline513_w": (dict(k=40.0, c=2.0, noise=2.0, seed=501), 502, 70),
}
PIPE_SCENES = {**PIPE257_SCENES, **PIPE513_SCENES}
AMBIGUOUS = 2e-6  # |value - 0.5| below this is fp32 evaluation noise: the decision may go either way
# Largest share of the lattice a fixture may leave undecided.  Measured: 0.00 / 0.00 / 0.06 % (257^3, f32
# noise), 0.02 % (configs[2], encoders in the loop), 0.01 % (513^3 f32), 0.4 % (513^3 under the f16-weight
# bound of 3e-4); round 5 allowed 25 %.
UNDECIDED_MAX = 0.01


def pipeline_res(name):
    return PIPE513_RES if name in PIPE513_SCENES else PIPE257_RES


def pipeline257_golden(name="pipeline257"):
    g = load_golden(name)
    rf = pipeline_res(name)[-1]
    queried = np.unpackbits(g["queried"])[:rf ** 3].astype(bool).reshape(rf, rf, rf)
    return g, queried


def pipeline257_inputs(name):
    head, feat_seed, step = PIPE_SCENES[name]
    return syn.body_mlp("G", **head), syn.body_feat(256, 128, 128, feat_seed), step


def pipeline257_undecided(g, queried, ambiguous=AMBIGUOUS, res=None):
    """Mask of the final lattice (257^3; 513^3 for the configs[4] fixture) where two fp32-class evaluations of the same field may legitimately
    take different octree decisions: the reach of every reference-queried node whose value is within
    AMBIGUOUS of the threshold.  A flip at level l (node spacing s_l) moves the boundary flags of the
    adjacent cells and, through the dilation boxes 9 / 7 / 3 / 3 (/ 3) of the finer levels, the selection
    within s_l + sum_{m>l} (box_m - 1) / 2 * s_m voxels (+ 2 for the interpolation footprint)."""
    res = list(res or PIPE257_RES)
    nl = len(res)
    rf = res[-1]
    assert queried.shape == (rf, rf, rf)
    spacing = [(rf - 1) // (r - 1) for r in res]  # 16, 8, 4, 2, 1
    box = [0, 9, 7] + [3] * (nl - 3)
    reach = [spacing[l] + sum((box[m] - 1) // 2 * spacing[m] for m in range(l + 1, nl)) + 2 for l in range(nl)]
    vals = np.zeros(queried.shape, np.float32)
    vals[queried] = g["values"]
    amb = np.argwhere(queried & (np.abs(vals - 0.5) <= ambiguous))
    mask = np.zeros(queried.shape, bool)
    for z, y, x in amb:
        level = next(l for l in range(nl) if z % spacing[l] == 0 and y % spacing[l] == 0 and x % spacing[l] == 0)
        r = reach[level]
        mask[max(z - r, 0):z + r + 1, max(y - r, 0):y + r + 1, max(x - r, 0):x + r + 1] = True
    return mask, len(amb)


def pipeline257_check(name, vol, queried, stats, tol, ambiguous=AMBIGUOUS):
    """Octree result (volume, queried-node mask or None, per-level counts) against the reference-driven
    fixture: outside the undecided regions the same nodes are queried and every queried value agrees
    within ``tol``; with no undecided node the per-level counts are equal too."""
    g, queried_ref = pipeline257_golden(name)
    undecided, n_amb = pipeline257_undecided(g, queried_ref, ambiguous, pipeline_res(name))
    firm = queried_ref & ~undecided
    ref_vol = np.zeros(queried_ref.shape, np.float32)
    ref_vol[queried_ref] = g["values"]
    err = float(np.abs(vol[firm] - ref_vol[firm]).max())
    frac = float(undecided.mean())
    print("%s: %d queried nodes, %d within %.0e of the threshold -> %.2f %% of the lattice undecided; "
          "max|value - reference| over the %d firm nodes = %.3g; margin %.3g"
          % (name, queried_ref.sum(), n_amb, ambiguous, 100 * frac, firm.sum(), err, float(g["margin"]) if "margin" in g else -1))
    assert frac <= UNDECIDED_MAX and err <= tol
    if queried is not None:
        assert np.array_equal(queried & ~undecided, firm)
    if n_amb == 0:
        assert list(stats) == list(g["stats"])
    else:  # the counts may differ by at most the nodes inside the undecided regions
        assert abs(sum(stats) - int(g["stats"].sum())) <= int(undecided.sum())
    return g, undecided, n_amb


def fixture_query_func(name):
    """query_func that answers from the fixture: the REFERENCE's value at every node it queried (NaN at any
    other node, which the callers assert never happens).  Driving oracle.seg3d_lossless with it rebuilds
    the complete volume the generator held (queried nodes exact, the rest interpolated by the schedule)."""
    g, queried = pipeline257_golden(name)
    rf = queried.shape[0]
    table = np.full(queried.shape, np.nan, np.float32)
    table[queried] = g["values"]

    def query_func(p):  # [3,N] world coordinates of lattice nodes: p = ((c + 0.5) / R) * 2 - 1
        c = np.rint((p.astype(np.float64) + 1.0) * 0.5 * rf - 0.5).astype(np.int64)
        out = table[c[2], c[1], c[0]]
        assert not np.isnan(out).any(), "the schedule asked for a node the reference run did not query"
        return out
    return query_func


def reference_driven_volume(oracle, name):
    """[R,R,R] f32 volume of the reference-driven run behind fixture ``name`` + its per-level counts."""
    stats = []
    vol = oracle.seg3d_lossless(fixture_query_func(name), [-1, -1, -1], [1, 1, 1], pipeline_res(name),
                                stats=stats)
    return vol, stats


@pytest.mark.parametrize("name", sorted(PIPE_SCENES))
def test_pipeline257_matches_reference(oracle, name):
    """BASELINE configs[1] size (and configs[4]'s 17..513): the octree driven by the fp32 C oracle takes the same
    decisions as when driven by the reference's netG.query (same queried node set, same per-level
    counts -- up to nodes the reference itself evaluated within fp32 noise of the threshold), the
    values agree to fp32 noise and forward_vertices gives the same columns."""
    g, queried_ref = pipeline257_golden(name)
    layers, f, step = pipeline257_inputs(name)
    calib = oracle.pifu_calib(*syn.scene_camera(step))
    assert np.array_equal(calib, g["calib"])
    stats = []
    queried = np.zeros_like(queried_ref)
    vol = oracle.seg3d_lossless(
        lambda p: oracle.query(f, p, calib[0], layers, 1, syn.Z_SCALE, precision="f32")[0],
        [-1, -1, -1], [1, 1, 1], pipeline_res(name), stats=stats, evaluated_out=queried)
    _, undecided, n_amb = pipeline257_check(name, vol, queried, stats, 5e-6)
    if name in PIPE513_SCENES:  # every node of the lattice, interpolated ones included, against the reference-driven volume
        ref_vol, ref_stats = reference_driven_volume(oracle, name)
        assert ref_stats == list(g["stats"])
        err = float(np.abs(vol - ref_vol)[~undecided].max())
        print("%s: max|oracle-driven - reference-driven| over all %d nodes outside the undecided reach = %.3g"
              % (name, int((~undecided).sum()), err))
        assert err <= 5e-6
    x, y, z, n = oracle.forward_vertices(vol, "front")
    if n_amb == 0:
        assert np.array_equal(x, g["X"].astype(np.int64)) and np.array_equal(y, g["Y"].astype(np.int64))
        assert np.abs(z - g["Z"]).max() <= 2e-3  # voxel units
    else:
        same = pipeline257_vertex_agreement(g, x, y, z)
        assert same >= 0.99


def pipeline257_vertex_agreement(g, x, y, z):
    """Fraction of the reference's visible vertices (X, Y) found at the same column with |dZ| <= 2e-3."""
    ref = {(int(a), int(b)): float(c) for a, b, c in zip(g["X"], g["Y"], g["Z"])}
    hit = sum(1 for a, b, c in zip(x, y, z) if abs(ref.get((int(a), int(b)), 1e9) - float(c)) <= 2e-3)
    return hit / max(len(ref), 1)


# ---- BASELINE configs[2]: both encoders in the loop (oracle/gen_golden.py: gen_pipeline257_color) ----
# name of the seeds: oracle/gen_golden.py COLOR257
COLOR257 = dict(img_g=75, img_c=76, enc_g=71, enc_c=72, head_g=dict(k=40.0, c=2.0, noise=0.05, seed=395),
                head_c=("rand", 77, 0.6), step=115)
# With an encoder in the loop two fp32 implementations differ by the encoder's rounding, not only the
# MLP's: features within ~6e-6 (tests/test_baseline_size_gpu.py), worth up to k * thick * 6e-6 / 4 ~
# 7e-6 on the occupancy through this head -- decisions on values closer to 0.5 may go either way.
COLOR257_AMBIGUOUS = 2e-5


def color257_nets(device="cpu"):
    """(netG, netC, fixture) of the configs[2] scene on ``device``: seeded encoders, the readout-body
    head fitted to the REFERENCE encoder's output (vector stored in the fixture), the netC head."""
    import torch
    from monoport_amd.modeling import PIFuNetC, PIFuNetG
    g = load_golden("pipeline257_color")
    cfg = COLOR257
    netg, netc = PIFuNetG().eval(), PIFuNetC().eval()
    for net, seed in ((netg, cfg["enc_g"]), (netc, cfg["enc_c"])):
        shapes = {k: tuple(v.shape) for k, v in net.image_filter.state_dict().items()}
        net.image_filter.load_state_dict(
            {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(shapes, seed).items()})
    heads = (syn.readout_body_mlp(g["readout"], float(g["r0"]), float(g["thick"]), **cfg["head_g"]),
             syn.rand_mlp("C", cfg["head_c"][1], cfg["head_c"][2]))
    for net, layers in zip((netg, netc), heads):
        sd = {}
        for i, (w, b) in enumerate(layers):
            sd["filters.%d.weight" % i] = torch.from_numpy(w)[:, :, None]
            sd["filters.%d.bias" % i] = torch.from_numpy(b)
        net.surface_classifier.load_state_dict(sd)
        net.to(device)
    return netg, netc, heads, g


def test_pipeline257_color_matches_reference(oracle):
    """BASELINE configs[2] end to end on the CPU: image -> our modules' torch-CPU encoders ->
    netC.filter(feat_prior) -> the 17..257 octree driven by the C oracle -> forward_vertices -> the
    colour chain, against the fixture the REFERENCE's modules produced for the same seeds (encoders,
    netG.query, forward_vertices, orthogonal, netC.query): same nodes, values, vertices, colours."""
    import torch
    netg, netc, (layers_g, layers_c), g = color257_nets("cpu")
    cfg = COLOR257
    with torch.no_grad():
        fg = netg.filter(torch.from_numpy(syn.synthetic_image(cfg["img_g"]))[None])
        fc = netc.filter(torch.from_numpy(syn.synthetic_image(cfg["img_c"]))[None], feat_prior=fg[-1][-1])
    assert np.abs(fg[-1][0][0, ::8, ::8, ::8].numpy() - g["featG_slice"]).max() <= 2e-5
    assert np.abs(fc[0][0][0, ::8, ::8, ::8].numpy() - g["featC_slice"]).max() <= 2e-5
    feat_g, feat_c = fg[-1][0][0].numpy(), fc[0][0][0].numpy()
    calib = oracle.pifu_calib(*syn.scene_camera(cfg["step"]))
    assert np.array_equal(calib, g["calib"])
    _, queried_ref = pipeline257_golden("pipeline257_color")
    stats, queried = [], np.zeros_like(queried_ref)
    vol = oracle.seg3d_lossless(
        lambda p: oracle.query(feat_g, p, calib[0], layers_g, 1, syn.Z_SCALE, precision="f32")[0],
        [-1, -1, -1], [1, 1, 1], PIPE257_RES, stats=stats, evaluated_out=queried)
    _, undecided, n_amb = pipeline257_check("pipeline257_color", vol, queried, stats, 2e-5, COLOR257_AMBIGUOUS)
    x, y, z, n = oracle.forward_vertices(vol, "front")
    same = pipeline257_vertex_agreement(g, x, y, z)
    assert same >= 0.999
    # colours on the REFERENCE's vertices: isolates the colour chain from the octree's coin flips
    mat = oracle.color_matrix([-1, -1, -1], [1, 1, 1], 257)

    def color_query(pts):
        return oracle.query(feat_c, pts, calib[0], layers_c, syn.LAST_OP["C"], syn.Z_SCALE, precision="f32")

    img = oracle.colorization(g["X"].astype(np.int64), g["Y"].astype(np.int64), g["Z"], 257,
                              color_query=color_query, mat_color=mat)
    err = float(np.abs(img - g["tex_image"]).max())
    print("configs[2] on the CPU: %.4f of the reference's %d vertices; max|colour - reference| = %.3g"
          % (same, g["X"].shape[0], err))
    assert err <= 1e-4
