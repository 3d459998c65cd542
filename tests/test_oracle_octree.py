"""Octree driver oracle (OUR restatement of implicit_seg.Seg3dLossless; parity unpinned -- the
dependency is not vendored).  Anchored on the property the upstream advertises: the
thresholded coarse-to-fine volume equals the thresholded dense evaluation, and every voxel the
octree queried carries the exact dense value.  CPU only."""
import numpy as np
import pytest

from monoport_amd import synthetic as syn

BMIN, BMAX = [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]


@pytest.fixture(scope="module")
def body_query(oracle):
    layers = syn.body_mlp("G", noise=0.05, seed=1)
    f = syn.body_feat(256, 128, 128, 2)
    ext, intr = syn.scene_camera(30)
    calib = oracle.pifu_calib(ext, intr)[0]
    return lambda pts: oracle.query(f, pts, calib, layers, 1, syn.Z_SCALE, precision="f32")[0]


def test_upsample2x_is_align_corners_trilinear(oracle):
    rs = np.random.RandomState(0)
    a = rs.rand(5, 5, 5).astype(np.float32)
    up = oracle.upsample2x(a)
    assert up.shape == (9, 9, 9)
    assert np.array_equal(up[::2, ::2, ::2], a)
    assert np.allclose(up[1, 0, 0], 0.5 * (a[0, 0, 0] + a[1, 0, 0]))
    assert np.allclose(up[1, 1, 1], a[:2, :2, :2].mean(), atol=1e-6)
    import torch
    ref = torch.nn.functional.interpolate(torch.from_numpy(a)[None, None], size=(9, 9, 9),
                                          mode="trilinear", align_corners=True)[0, 0].numpy()
    # same interpolant; torch sums the 8 weighted corners in another order (<= 1 ulp apart)
    assert np.abs(up - ref).max() <= 1.2e-7


def test_dilate_box_matches_conv(oracle):
    import torch
    rs = np.random.RandomState(1)
    m = rs.rand(12, 12, 12) > 0.97
    for k in (3, 7, 9):
        ref = torch.nn.functional.conv3d(torch.from_numpy(m.astype(np.float32))[None, None],
                                         torch.ones(1, 1, k, k, k), padding=k // 2)[0, 0] > 0
        assert np.array_equal(oracle.dilate_box(m, k), ref.numpy())


def test_lattice_points_convention(oracle):
    idx = np.array([[0, 0, 0], [16, 8, 4]])
    p = oracle.lattice_points(idx, 16, 257, BMIN, BMAX)
    assert p.shape == (3, 2) and p.dtype == np.float32
    assert np.allclose(p[:, 0], (0.5 / 257) * 2 - 1, atol=1e-6)
    # row is (z, y, x) -> world (x, y, z)
    assert np.allclose(p[:, 1], [((4 * 16 + 0.5) / 257) * 2 - 1, ((8 * 16 + 0.5) / 257) * 2 - 1,
                                 ((16 * 16 + 0.5) / 257) * 2 - 1], atol=1e-6)


@pytest.mark.parametrize("res", [[9, 17, 33], [5, 9, 17, 33]])
def test_lossless_vs_dense(oracle, body_query, res):
    stats = []
    vol = oracle.seg3d_lossless(body_query, BMIN, BMAX, res, stats=stats)
    dense = oracle.dense_volume(body_query, BMIN, BMAX, res[-1])
    assert vol.shape == dense.shape == (res[-1],) * 3
    assert np.array_equal(vol > 0.5, dense > 0.5)
    near = np.abs(dense - 0.5) < 0.3  # near-surface voxels were queried exactly
    assert np.array_equal(vol[near], dense[near])
    assert sum(stats) < 0.5 * res[-1] ** 3


@pytest.mark.parametrize("res", [[9, 17, 33], [5, 9, 17, 33, 65]])
def test_non_faster_mode_is_lossless_and_re_examines(oracle, body_query, res):
    """faster=False: 3^3 dilation + conflict loop.  Still lossless against dense evaluation; it
    queries fewer points at the coarse levels than the 9^3 / 7^3 schedule and recovers by
    re-examining conflicts (at least one round on this body)."""
    stats, rounds, stats_fast = [], [], []
    vol = oracle.seg3d_lossless(body_query, BMIN, BMAX, res, stats=stats, faster=False,
                                rounds=rounds)
    oracle.seg3d_lossless(body_query, BMIN, BMAX, res, stats=stats_fast)
    dense = oracle.dense_volume(body_query, BMIN, BMAX, res[-1])
    assert np.array_equal(vol > 0.5, dense > 0.5)
    assert len(stats) == len(rounds) == len(res) and rounds[0] == 0
    assert sum(rounds) >= 1 and stats[1] < stats_fast[1]


def test_empty_returns_none(oracle):
    assert oracle.seg3d_lossless(lambda p: np.zeros(p.shape[1], np.float32), BMIN, BMAX,
                                 [5, 9]) is None


def test_rejects_non_doubling(oracle):
    with pytest.raises(ValueError):
        oracle.seg3d_lossless(lambda p: np.zeros(p.shape[1], np.float32), BMIN, BMAX, [5, 11])


def test_final_level_rules(oracle, body_query):
    """The three selection rules of the LAST level (Seg3dLossless(final_level=...), mp_recon_batch_ex):
    "dilate3" is the lossless schedule; "upstream" (nodes whose upsampled inside-mask is exactly 0.5,
    undilated) evaluates a subset of its nodes -- several times fewer -- and "interpolate" none; the
    coarser levels are the same for all three, and the price of the cheaper rules is stated against
    dense evaluation (measured here on the 65^3 body: the numbers are printed)."""
    res = [5, 9, 17, 33, 65]
    dense = oracle.dense_volume(body_query, BMIN, BMAX, res[-1])
    out = {}
    for rule in oracle.FINAL_LEVELS:
        stats, ev = [], np.zeros((res[-1],) * 3, bool)
        vol = oracle.seg3d_lossless(body_query, BMIN, BMAX, res, stats=stats, evaluated_out=ev, final_level=rule)
        wrong = int(((vol > 0.5) != (dense > 0.5)).sum())
        out[rule] = (vol, stats, ev, wrong)
        print("final_level=%-11s points per level %s, %d of %d inside voxels differ from dense evaluation"
              % (rule, stats, wrong, int((dense > 0.5).sum())))
    d3, up, ip = out["dilate3"], out["upstream"], out["interpolate"]
    assert d3[3] == 0                                             # lossless
    assert d3[1][:-1] == up[1][:-1] == ip[1][:-1]                 # the same coarse levels
    assert ip[1][-1] == 0 and 0 < up[1][-1] < 0.5 * d3[1][-1]     # none / a fraction of the nodes
    assert not (up[2] & ~d3[2]).any()                             # a subset of the lossless rule's nodes
    assert np.array_equal(up[0][up[2]], dense[up[2]])             # what was evaluated is exact
    # the price: 2 % of the inside voxels at 65^3 (0.4 % at 257^3, tests/test_recon_gpu.py), more without any evaluation
    assert ip[3] >= up[3] > 0 and up[3] <= 0.05 * int((dense > 0.5).sum())
    with pytest.raises(ValueError):
        oracle.seg3d_lossless(body_query, BMIN, BMAX, res, final_level="upstream", faster=False)
