"""Parity of the fused HIP query kernel (through the C-ABI) with the CPU oracle and with the
golden vectors produced by the reference itself.  Needs an MI355X."""
import numpy as np
import pytest

from conftest import load_golden
from monoport_amd import synthetic as syn
from test_oracle_golden import QUERY_CASES, query_inputs

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL_REF = 1e-4   # north-star bar against the reference's own output
TOL_ORACLE = 2e-5  # against our fp32 oracle (same op order for everything but the GEMM sums)


@pytest.fixture(scope="module")
def ops():
    from monoport_amd import ops as _ops
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return _ops


def _gpu_query(ops, kind, layers, feat, pts, calib):
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, syn.LAST_OP[kind])
    f = ops.pack_features(torch.from_numpy(feat)[None].to(dev))
    out = ops.query(mlp, f, torch.from_numpy(pts)[None].to(dev), torch.from_numpy(calib).to(dev),
                    syn.Z_SCALE)
    torch.cuda.synchronize()
    return out[0].cpu().numpy()


@pytest.mark.parametrize("name", sorted(QUERY_CASES))
def test_query_vs_reference_golden_and_oracle(ops, oracle, name):
    g = load_golden(name)
    kind, layers, f, p = query_inputs(name)
    out = _gpu_query(ops, kind, layers, f, p, g["calib"])
    assert out.shape == g["out"].shape
    assert np.isfinite(out).all()
    assert np.abs(out - g["out"]).max() <= TOL_REF
    ref32 = oracle.query(f, p, g["calib"][0], layers, syn.LAST_OP[kind], syn.Z_SCALE,
                         precision="f32")
    assert np.abs(out - ref32).max() <= 3 * TOL_ORACLE
    # accuracy against the fp64 oracle: the HIP kernel must be no noisier than the reference's own
    # fp32 evaluation (whose error vs fp64 is 3.1e-5 on the netC fixture, 4e-6 on netG)
    ref64 = oracle.query(f, p, g["calib"][0], layers, syn.LAST_OP[kind], syn.Z_SCALE,
                         precision="f64")
    err_gpu = np.abs(out - ref64).max()
    err_ref = np.abs(g["out"] - ref64).max()
    print("%s: |gpu-f64| %.3g  |reference-f64| %.3g" % (name, err_gpu, err_ref))
    assert err_gpu <= max(2 * err_ref, 1e-5)
    # out-of-image points are exactly 0 (MonoPortNet.py:89)
    xyz = oracle.orthogonal(p, g["calib"][0])
    outside = np.minimum(1 - np.abs(xyz[0]), 1 - np.abs(xyz[1])) < -1e-6
    assert (out[:, outside] == 0).all()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 1000, 4097])
def test_query_ragged_sizes(ops, oracle, n):
    layers = syn.rand_mlp("G", 5, 2.0)
    f = syn.rand_feat(256, 128, 128, 6)
    p = syn.rand_points(n, 100 + n, 1.1)
    calib = oracle.pifu_calib(*syn.scene_camera(40))
    out = _gpu_query(ops, "G", layers, f, p, calib)
    ref = oracle.query(f, p, calib[0], layers, 1, syn.Z_SCALE, precision="f32")
    assert out.shape == (1, n)
    assert np.abs(out - ref).max() <= TOL_ORACLE


def test_query_empty(ops):
    layers = syn.rand_mlp("G", 5, 1.0)
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, 1)
    f = ops.pack_features(torch.zeros(1, 256, 128, 128, device=dev))
    out = ops.query(mlp, f, torch.zeros(1, 3, 0, device=dev), torch.eye(4, device=dev)[None], 1.28)
    assert out.shape == (1, 1, 0)


def test_query_strided_points_and_small_map(ops, oracle):
    """query_func hands netG.query a permuted [1,N,3] view (RTL/main.py:176-177)."""
    layers = syn.rand_mlp("G", 7, 1.5)
    f = syn.rand_feat(256, 40, 24, 8)  # non-square, non-128 map
    p = syn.rand_points(777, 9, 1.0)
    calib = oracle.pifu_calib(*syn.scene_camera(10))
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, 1)
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    pts_n3 = torch.from_numpy(np.ascontiguousarray(p.T))[None].to(dev)  # [1,N,3]
    out = ops.query(mlp, fh, pts_n3.permute(0, 2, 1), torch.from_numpy(calib).to(dev), syn.Z_SCALE)
    ref = oracle.query(f, p, calib[0], layers, 1, syn.Z_SCALE, precision="f32")
    assert np.abs(out[0].cpu().numpy() - ref).max() <= TOL_ORACLE


def test_query_identity_calib_all_in_image(ops, oracle):
    layers = syn.rand_mlp("C", 17, 1.5)
    f = syn.rand_feat(512, 128, 128, 18)
    p = syn.rand_points(3000, 19, 1.0)
    p[:, :4] = np.array([[-1, 1, -1, 1], [-1, -1, 1, 1], [0, 0, 0, 0]], np.float32)  # corners
    calib = np.eye(4, dtype=np.float32)[None]
    out = _gpu_query(ops, "C", layers, f, p, calib)
    ref = oracle.query(f, p, calib[0], layers, 2, syn.Z_SCALE, precision="f32")
    assert np.abs(out - ref).max() <= TOL_ORACLE
    assert (np.abs(out) > 0).all()


def test_index_vs_reference(ops):
    g = load_golden("index")
    f = syn.rand_feat(256, 128, 128, 41)
    dev = "cuda:0"
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    out = ops.index(fh, torch.from_numpy(g["uv"])[None].to(dev))[0].cpu().numpy()
    assert np.array_equal(out, g["out"])  # the reference's own FMA chain: identical bits


def test_orthogonal_vs_reference(ops):
    g = load_golden("orthogonal")
    p = syn.rand_points(1000, 43, 1.0)
    dev = "cuda:0"
    out = ops.orthogonal(torch.from_numpy(p)[None].to(dev), torch.from_numpy(g["calib"]).to(dev))
    assert np.array_equal(out[0].cpu().numpy(), g["out"])  # torch.baddbmm bits (MKL FMA chain)


def test_pack_features_concat(ops):
    dev = "cuda:0"
    a = torch.randn(1, 256, 16, 24, device=dev)
    b = torch.randn(1, 256, 16, 24, device=dev)
    out = ops.pack_features([a, b])
    ref = torch.cat([a, b], 1)[0].permute(1, 2, 0).contiguous()
    assert torch.equal(out, ref)


def test_unsupported_shapes_fail_loudly(ops):
    from monoport_amd._lib import MonoportError
    with pytest.raises(MonoportError):
        ops.PackedMLP(ops.get_context("cuda:0"), [129, 1024, 512, 256, 128, 1], 1)
    with pytest.raises(MonoportError):
        ops.get_context("cpu")


@pytest.mark.parametrize("kind,n", [("G", 5000), ("C", 777), ("G", 64), ("G", 1)])
def test_surface_classifier_forward_on_explicit_features(ops, oracle, kind, n):
    """SurfaceClassifier.forward (the reference's __main__ micro-benchmark shape,
    heads/SurfaceClassifier.py:95-116) against a float64 numpy restatement of :47-69."""
    from monoport_amd.modeling.heads import PIFuNetCMLP, PIFuNetGMLP
    layers = syn.rand_mlp(kind, 23, 2.0)
    head = (PIFuNetGMLP if kind == "G" else PIFuNetCMLP)()
    head.load_state_dict({**{"filters.%d.weight" % i: torch.from_numpy(w)[:, :, None] for i, (w, _) in enumerate(layers)},
                          **{"filters.%d.bias" % i: torch.from_numpy(b) for i, (_, b) in enumerate(layers)}})
    head.to("cuda:0").eval()
    rs = np.random.RandomState(n)
    x = rs.standard_normal((syn.MLP_DIMS[kind][0], n)).astype(np.float32)
    out = head(torch.from_numpy(x)[None].to("cuda:0"))[0].cpu().numpy()
    y = x.astype(np.float64)
    x64 = y
    for i, (w, b) in enumerate(layers):
        inp = y if i == 0 else np.concatenate([y, x64], 0)  # SurfaceClassifier.py:55
        y = w.astype(np.float64) @ inp + b.astype(np.float64)[:, None]
        if i != len(layers) - 1:
            y = np.where(y > 0, y, 0.01 * y)
    ref = 1 / (1 + np.exp(-y)) if kind == "G" else np.tanh(y)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 2e-5


@pytest.mark.parametrize("name", ["query_G_rand", "query_G_body"])
def test_f16x3_query_is_f32_class(ops, oracle, name):
    """The split-precision kernel (three f16 MFMAs per product, f32 accumulate) must meet the same
    bars as the f32 kernel: 1e-4 against the reference's golden output, no noisier than the
    reference's own fp32 evaluation against the fp64 oracle, and within 2e-6 of the f32 kernel."""
    g = load_golden(name)
    kind, layers, f, p = query_inputs(name)
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, syn.LAST_OP[kind])
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    pts = torch.from_numpy(p)[None].to(dev)
    cal = torch.from_numpy(g["calib"]).to(dev)
    out32 = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)[0].cpu().numpy()
    mlp.set_precision("f16x3")
    out16 = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)[0].cpu().numpy()
    mlp.set_precision("f32")
    again = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)[0].cpu().numpy()
    assert np.array_equal(again, out32)  # switching back restores the exact f32 path
    assert np.isfinite(out16).all()
    assert np.abs(out16 - g["out"]).max() <= TOL_REF
    ref64 = oracle.query(f, p, g["calib"][0], layers, syn.LAST_OP[kind], syn.Z_SCALE, precision="f64")
    err16, err_ref = np.abs(out16 - ref64).max(), np.abs(g["out"] - ref64).max()
    print("%s f16x3: |gpu-f64| %.3g  |reference-f64| %.3g  |f16x3-f32 kernel| %.3g"
          % (name, err16, err_ref, np.abs(out16 - out32).max()))
    assert err16 <= max(2 * err_ref, 1e-5)
    assert np.abs(out16 - out32).max() <= 2e-6
    xyz = oracle.orthogonal(p, g["calib"][0])
    outside = np.minimum(1 - np.abs(xyz[0]), 1 - np.abs(xyz[1])) < -1e-6
    assert (out16[:, outside] == 0).all()


@pytest.mark.parametrize("n", [1, 127, 128, 129, 1000])
def test_f16x3_ragged_sizes_and_large_weights(ops, oracle, n):
    """128-point tiles: ragged tails; weights spanning 1e-3..40 exercise the per-layer scaling."""
    layers = syn.body_mlp("G", noise=0.3, seed=n)  # entries from ~1e-3 up to k = 40
    f = syn.body_feat(256, 128, 128, 6)
    p = syn.rand_points(n, 200 + n, 1.05)
    calib = oracle.pifu_calib(*syn.scene_camera(40))
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, 1)
    mlp.set_precision("f16x3")
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    out = ops.query(mlp, fh, torch.from_numpy(p)[None].to(dev), torch.from_numpy(calib).to(dev),
                    syn.Z_SCALE)[0].cpu().numpy()
    ref = oracle.query(f, p, calib[0], layers, 1, syn.Z_SCALE, precision="f64")
    assert out.shape == (1, n) and np.abs(out - ref).max() <= 2e-5


def test_f16x3_rejects_netc_head(ops):
    from monoport_amd._lib import MonoportError
    mlp = ops.PackedMLP.from_layers("cuda:0", syn.rand_mlp("C", 3, 1.0), 2)
    with pytest.raises(MonoportError):
        mlp.set_precision("f16x3")


@pytest.mark.parametrize("calib_kind", ["scene", "identity"])
def test_config1_dense_lattice_64(ops, oracle, calib_kind):
    """BASELINE configs[0]: the dense 64^3 lattice p = ((i+0.5)/64)*2-1,
    z-major, 262,144 points through one query call, against the CPU oracle."""
    r = 64
    layers = syn.body_mlp("G", noise=0.05, seed=1)
    f = syn.body_feat(256, 128, 128, 2)
    calib = (oracle.pifu_calib(*syn.scene_camera(20)) if calib_kind == "scene"
             else np.eye(4, dtype=np.float32)[None])
    c = (np.arange(r, dtype=np.float32) + 0.5) / r * 2 - 1
    zz, yy, xx = np.meshgrid(c, c, c, indexing="ij")
    p = np.stack([xx.ravel(), yy.ravel(), zz.ravel()]).astype(np.float32)  # [3, 64^3], [z,y,x] order
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, 1)
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    out = ops.query(mlp, fh, torch.from_numpy(p)[None].to(dev), torch.from_numpy(calib).to(dev),
                    syn.Z_SCALE)[0].cpu().numpy()
    ref = oracle.query(f, p, calib[0], layers, 1, syn.Z_SCALE, precision="f32")
    assert out.shape == (1, r ** 3)
    assert np.abs(out - ref).max() <= 2e-5
    inside = (out > 0.5).mean()
    assert 0.01 < inside < 0.3  # the analytic body is a closed blob, not a degenerate field


@pytest.mark.parametrize("precision,tol", [("f16w", 3e-4), ("f16", 5e-3)])
@pytest.mark.parametrize("name", ["query_G_rand", "query_G_body"])
def test_fp16_modes_report_error(ops, oracle, name, precision, tol):
    """BASELINE configs[4] arithmetic ("fp16 weights"): NOT held to the 1e-4 bar (SURVEY.md appendix A
    item 10 measured 7.1e-5 for weight rounding alone on the reference); the bound asserted here
    is the looser one written above and the measured error is printed."""
    g = load_golden(name)
    kind, layers, f, p = query_inputs(name)
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, syn.LAST_OP[kind])
    mlp.set_precision(precision)
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    out = ops.query(mlp, fh, torch.from_numpy(p)[None].to(dev), torch.from_numpy(g["calib"]).to(dev),
                    syn.Z_SCALE)[0].cpu().numpy()
    ref64 = oracle.query(f, p, g["calib"][0], layers, syn.LAST_OP[kind], syn.Z_SCALE, precision="f64")
    err = np.abs(out - ref64).max()
    print("%s %s: |gpu - f64 oracle| max %.3g" % (name, precision, err))
    assert np.isfinite(out).all() and err <= tol


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15])
def test_f16x3_no_noisier_than_f32_kernel_on_random_heads(ops, oracle, seed):
    """Random-weight heads (gain 1-3), random features and cameras: against the fp64 oracle the
    split-precision kernel's error stays within 1.5x of the exact-f32 kernel's own error."""
    rs = np.random.RandomState(seed)
    layers = syn.rand_mlp("G", seed, float(rs.uniform(1.0, 3.0)))
    f = syn.rand_feat(256, 128, 128, seed)
    p = syn.rand_points(3000, seed, 1.1)
    calib = oracle.pifu_calib(*syn.scene_camera(int(rs.randint(0, 120))))
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, 1)
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    pts = torch.from_numpy(p)[None].to(dev)
    cal = torch.from_numpy(calib).to(dev)
    out32 = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)[0].cpu().numpy()
    mlp.set_precision("f16x3")
    out16 = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)[0].cpu().numpy()
    ref = oracle.query(f, p, calib[0], layers, 1, syn.Z_SCALE, precision="f64")
    e32, e16 = np.abs(out32 - ref).max(), np.abs(out16 - ref).max()
    print("seed %d: |f32 kernel - f64| %.3g, |f16x3 - f64| %.3g" % (seed, e32, e16))
    assert e32 <= 1e-4 and e16 <= 1e-4 and e16 <= 1.5 * e32 + 1e-7


def _tuned(mode):
    """Context manager around mp_query_tune (0 = 64-point tiles, 1 = 32-point tiles, -1 = default gate at 2048 tiles)."""
    import contextlib
    from monoport_amd import _lib

    @contextlib.contextmanager
    def cm():
        lib = _lib.load()
        lib.mp_query_tune(mode)
        try:
            yield
        finally:
            lib.mp_query_tune(-1)
    return cm()


@pytest.mark.parametrize("cout", [1, 3])
@pytest.mark.parametrize("n", [1, 31, 32, 33, 95, 1000, 40000])
def test_small_tile_kernel_is_bit_identical(ops, n, cout):
    """query_small.hip (32-point tiles, for launches with few tiles) returns the bits of
    the 64-point kernel of query.hip: same K order, same FMA chains, bias first -- the parity
    statements of this file hold for both."""
    layers = syn.rand_mlp("G", 31 + cout, 2.0)
    if cout == 3:  # a 3-channel head on a 256-channel map (the C = 256, Cout = 3 instantiation)
        rs = np.random.RandomState(5)
        w4, b4 = layers[-1]
        layers[-1] = (rs.uniform(-0.1, 0.1, (3, w4.shape[1])).astype(np.float32),
                      rs.uniform(-0.1, 0.1, (3,)).astype(np.float32))
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, 1 if cout == 1 else 2)
    f = ops.pack_features(torch.from_numpy(syn.rand_feat(256, 128, 128, 6))[None].to(dev))
    p = torch.from_numpy(syn.rand_points(n, 100 + n, 1.1))[None].to(dev)
    cal = torch.from_numpy(np.eye(4, dtype=np.float32)[None]).to(dev)
    cal[0, 0, 0] = 0.93
    outs = {}
    for mode in (0, 1, -1):
        with _tuned(mode):
            outs[mode] = ops.query(mlp, f, p, cal, syn.Z_SCALE)
    assert outs[0].shape == (1, cout, n)
    assert float(outs[0].abs().max()) > 0
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], outs[-1])


def test_small_tile_kernel_device_counts(ops):
    """With the counts on the device both kernels are launched and each reads the counts: exactly
    one of them produces the frame set, on either side of the 2048-tile gate (ragged, empty and full
    frames), and the results equal the forced single-kernel runs."""
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, syn.rand_mlp("G", 61, 2.0), 1)
    cap = 40000  # 625 tiles of capacity per frame: 4-5 frames are above the gate, so the choice is made on the device
    for counts_host in ([4097, 0, 64, 5000, 1], [40000, 13, 0, 977, 64], [40000] * 5,
                        [40000, 40000, 40000, 172 * 64], [40000, 40000, 40000, 172 * 64 + 1],
                        [40000, 40000, 40000, 171 * 64 + 1, 1]):
        feats, pts, cnts, cals = [], [], [], []
        for i, c in enumerate(counts_host):
            feats.append(ops.pack_features(torch.from_numpy(syn.rand_feat(256, 64, 64, 70 + i))[None].to(dev)))
            pts.append(torch.from_numpy(syn.rand_points(cap, 80 + i, 1.0)).to(dev).contiguous())
            cnts.append(torch.tensor([c], dtype=torch.int32, device=dev))
            cal = np.eye(4, dtype=np.float32)[None]
            cal[0, 1, 1] = 1.0 - 0.04 * i
            cals.append(torch.from_numpy(cal).to(dev))
        outs = {}
        for mode in (0, 1, -1):
            with _tuned(mode):
                outs[mode] = ops.query_counted_batch(mlp, feats, pts, cnts, cals, syn.Z_SCALE)
        for i, c in enumerate(counts_host):
            assert torch.equal(outs[0][i], outs[1][i]), (counts_host, i)
            assert torch.equal(outs[0][i], outs[-1][i]), (counts_host, i)
            if c:
                assert float(outs[-1][i][:, :c].abs().max()) > 0.0
            if c < cap:
                assert float(outs[-1][i][:, c:].abs().max()) == 0.0


@pytest.mark.parametrize("cout", [1, 3])
def test_skip_table_matches_float64_products(ops, oracle, cout):
    """mp_skip_table: table[y,x,:] = the feature-segment weights of layers 0-4 times feat[y,x,:],
    against float64 products (rows 0 / 1024 / 1536 / 1792 / 1920)."""
    layers = syn.rand_mlp("G", 77, 2.0)
    if cout == 3:
        rs = np.random.RandomState(5)
        w4 = layers[-1][0]
        layers[-1] = (rs.uniform(-0.1, 0.1, (3, w4.shape[1])).astype(np.float32),
                      rs.uniform(-0.1, 0.1, (3,)).astype(np.float32))
    f = syn.rand_feat(256, 40, 24, 8)  # 960 texels = 15 tiles of 64
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, 1 if cout == 1 else 2)
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    try:
        table = ops.skip_table(mlp, fh)
        torch.cuda.synchronize()
        got = table.table.cpu().numpy()
        assert got.shape == (40, 24, ops.SKIP_TABLE_ROWS)
        row0 = 0
        for l, (w, _) in enumerate(layers):
            hidden = w.shape[1] - 257  # input = [hidden | 256 features | z]
            wx = w[:, hidden:hidden + 256].astype(np.float64)
            ref = np.einsum("rc,chw->hwr", wx, f.astype(np.float64))
            part = got[:, :, row0:row0 + w.shape[0]]
            assert np.abs(part - ref).max() <= 2e-6 * np.abs(ref).max(), l
            row0 += w.shape[0]
        assert row0 == 1920 + cout
    finally:
        ops.skip_table_release(mlp.ctx)


@pytest.mark.parametrize("name", ["query_G_rand", "query_G_body"])
def test_skip_table_query_vs_reference_golden_and_plain_path(ops, oracle, name):
    """The query through the skip table (feature-segment weights applied per texel, four rows blended per point)
    against the reference's golden output, the fp32 / fp64 oracle and the plain fused kernel: the
    same field up to f32 rounding, exact zeros outside the image, no noisier than the reference."""
    g = load_golden(name)
    kind, layers, f, p = query_inputs(name)
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, syn.LAST_OP[kind])
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    pts = torch.from_numpy(p)[None].to(dev)
    cal = torch.from_numpy(g["calib"]).to(dev)
    plain = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)[0].cpu().numpy()
    try:
        table = ops.skip_table(mlp, fh)
        out = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)[0].cpu().numpy()
        # an unrelated map without a table in the same context still runs the plain path
        fh2 = fh.clone()
        assert torch.equal(ops.query(mlp, fh2, pts, cal, syn.Z_SCALE)[0].cpu(), torch.from_numpy(plain))
    finally:
        ops.skip_table_release(mlp.ctx)
    again = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)[0].cpu().numpy()
    assert np.array_equal(again, plain)  # released: back on the plain path
    assert not np.array_equal(out, plain)  # ... and the table path really ran
    assert np.abs(out - g["out"]).max() <= TOL_REF
    ref32 = oracle.query(f, p, g["calib"][0], layers, syn.LAST_OP[kind], syn.Z_SCALE, precision="f32")
    assert np.abs(out - ref32).max() <= 3 * TOL_ORACLE
    ref64 = oracle.query(f, p, g["calib"][0], layers, syn.LAST_OP[kind], syn.Z_SCALE, precision="f64")
    err_tab, err_plain, err_ref = (np.abs(v - ref64).max() for v in (out, plain, g["out"]))
    print("%s: |table-f64| %.3g  |plain-f64| %.3g  |reference-f64| %.3g  |table-plain| %.3g"
          % (name, err_tab, err_plain, err_ref, np.abs(out - plain).max()))
    assert err_tab <= max(2 * err_ref, 1e-5)
    xyz = oracle.orthogonal(p, g["calib"][0])
    outside = np.minimum(1 - np.abs(xyz[0]), 1 - np.abs(xyz[1])) < -1e-6
    assert (out[:, outside] == 0).all()
    del table


@pytest.mark.parametrize("n", [1, 31, 33, 1000, 40000])
def test_skip_table_ragged_sizes_and_device_counts(ops, oracle, n):
    layers = syn.rand_mlp("G", 5, 2.0)
    f = syn.rand_feat(256, 64, 64, 6)
    p = syn.rand_points(n, 100 + n, 1.1)
    calib = oracle.pifu_calib(*syn.scene_camera(40))
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, 1)
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    try:
        table = ops.skip_table(mlp, fh)
        out = ops.query(mlp, fh, torch.from_numpy(p)[None].to(dev), torch.from_numpy(calib).to(dev), syn.Z_SCALE)
        cap = n + 7
        pts = torch.zeros((3, cap), device=dev)
        pts[:, :n] = torch.from_numpy(p).to(dev)
        cnt = torch.tensor([n], dtype=torch.int32, device=dev)
        counted = ops.query_counted(mlp, fh, pts.contiguous(), cnt, torch.from_numpy(calib).to(dev), syn.Z_SCALE)
    finally:
        ops.skip_table_release(mlp.ctx)
    ref = oracle.query(f, p, calib[0], layers, 1, syn.Z_SCALE, precision="f32")
    assert out.shape == (1, 1, n)
    assert np.abs(out[0].cpu().numpy() - ref).max() <= TOL_ORACLE
    assert torch.equal(counted[:, :n], out[0]) and float(counted[:, n:].abs().max()) == 0.0
    del table


def test_skip_table_registry_semantics(ops, oracle):
    """Registration follows the feature map: a handle unregisters only what is still its own, a
    rewritten map needs a new table (the old one would be stale), netC heads (C = 512) and
    odd-sized maps are refused, and destroying a head drops its tables."""
    dev = "cuda:0"
    layers = syn.rand_mlp("G", 9, 2.0)
    mlp = ops.PackedMLP.from_layers(dev, layers, 1)
    f1, f2 = syn.rand_feat(256, 64, 64, 1), syn.rand_feat(256, 64, 64, 2)
    fh = ops.pack_features(torch.from_numpy(f1)[None].to(dev))
    p = syn.rand_points(5000, 4, 1.0)
    pts = torch.from_numpy(p)[None].to(dev)
    cal = torch.from_numpy(np.eye(4, dtype=np.float32)[None]).to(dev)
    plain1 = ops.query(mlp, fh, pts, cal, syn.Z_SCALE).clone()
    buf = torch.empty((64, 64, ops.SKIP_TABLE_ROWS), device=dev)
    h1 = ops.skip_table(mlp, fh, out=buf)
    tab1 = ops.query(mlp, fh, pts, cal, syn.Z_SCALE).clone()
    assert not torch.equal(tab1, plain1) and float((tab1 - plain1).abs().max()) <= 2e-6
    # the map is rewritten in place: a new table into the same buffer supersedes the old handle
    fh.copy_(ops.pack_features(torch.from_numpy(f2)[None].to(dev)))
    h2 = ops.skip_table(mlp, fh, out=buf)
    h1.release()  # must NOT unregister h2's registration of the same (map, table) pair
    tab2 = ops.query(mlp, fh, pts, cal, syn.Z_SCALE).clone()
    ref2 = oracle.query(f2, p, np.eye(4, dtype=np.float32), layers, 1, syn.Z_SCALE, precision="f32")
    assert np.abs(tab2[0].cpu().numpy() - ref2).max() <= TOL_ORACLE
    h2.release()
    plain2 = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)
    assert not torch.equal(plain2, tab2) and float((tab2 - plain2).abs().max()) <= 2e-6
    # a second head on the same map: its launches do not pick up the first head's table
    mlp_b = ops.PackedMLP.from_layers(dev, syn.rand_mlp("G", 10, 2.0), 1)
    h3 = ops.skip_table(mlp, fh)
    out_b = ops.query(mlp_b, fh, pts, cal, syn.Z_SCALE)
    h3.release()
    assert torch.equal(out_b, ops.query(mlp_b, fh, pts, cal, syn.Z_SCALE))
    # refused: netC head (C = 512), texel count not a multiple of 64
    from monoport_amd._lib import MonoportError
    mlp_c = ops.PackedMLP.from_layers(dev, syn.rand_mlp("C", 3, 1.0), 2)
    with pytest.raises(MonoportError, match="skip table"):
        ops.skip_table(mlp_c, ops.pack_features(torch.zeros(1, 512, 64, 64, device=dev)))
    with pytest.raises(MonoportError, match="skip table"):
        ops.skip_table(mlp, ops.pack_features(torch.zeros(1, 256, 10, 10, device=dev)))


@pytest.mark.parametrize("precision,tol64", [("f16x3", None), ("f16w", 3e-4), ("f16", 5e-3)])
@pytest.mark.parametrize("name", ["query_G_rand", "query_G_body"])
def test_f16_kernels_through_the_skip_table(ops, oracle, name, precision, tol64, monkeypatch):
    """Round 4: the split-precision kernels blend the (exact f32) skip table too (pifu_query16_tab_kernel):
    layer 0 and the skip rows come from the table, the hidden GEMMs run on f16 MFMA as before.  f16x3 is
    held to the f32 bars (1e-4 against the reference, no noisier than the reference's own fp32 evaluation,
    within 2e-6 of the exact-f32 table path); f16w / f16 to the looser bounds of their plain kernels --
    and they may only get BETTER, since 42 % of the products are now exact f32.  (Only f16x3 takes this
    kernel by default -- it is the one precision that gets faster; MONOPORT_TAB16=all routes the others.)"""
    monkeypatch.setenv("MONOPORT_TAB16", "all")
    g = load_golden(name)
    kind, layers, f, p = query_inputs(name)
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, syn.LAST_OP[kind])
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    pts = torch.from_numpy(p)[None].to(dev)
    cal = torch.from_numpy(g["calib"]).to(dev)
    ref64 = oracle.query(f, p, g["calib"][0], layers, syn.LAST_OP[kind], syn.Z_SCALE, precision="f64")
    mlp.set_precision(precision)
    plain16 = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)[0].cpu().numpy()
    try:
        table = ops.skip_table(mlp, fh)
        out = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)[0].cpu().numpy()
        for n in (1, 95, 96, 97, 1000):  # ragged tails of the 96-point tile
            part = ops.query(mlp, fh, pts[:, :, :n].contiguous(), cal, syn.Z_SCALE)[0].cpu().numpy()
            assert np.array_equal(part, out[:, :n]), n
        mlp.set_precision("f32")
        tab32 = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)[0].cpu().numpy()
    finally:
        ops.skip_table_release(mlp.ctx)
    assert np.isfinite(out).all() and not np.array_equal(out, plain16)  # the table kernel ran
    err, err_plain, err_ref = np.abs(out - ref64).max(), np.abs(plain16 - ref64).max(), np.abs(g["out"] - ref64).max()
    print("%s %s through the table: |gpu-f64| %.3g (plain %s kernel %.3g, reference %.3g), |%s table - f32 table| %.3g"
          % (name, precision, err, precision, err_plain, err_ref, precision, np.abs(out - tab32).max()))
    if precision == "f16x3":
        assert np.abs(out - g["out"]).max() <= TOL_REF
        assert err <= max(2 * err_ref, 1e-5) and np.abs(out - tab32).max() <= 2e-6
    else:
        assert err <= tol64 and err <= 1.5 * err_plain + 1e-6
    xyz = oracle.orthogonal(p, g["calib"][0])
    outside = np.minimum(1 - np.abs(xyz[0]), 1 - np.abs(xyz[1])) < -1e-6
    assert (out[:, outside] == 0).all()
    del table


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_three_output_head_through_the_skip_table(ops, oracle, precision):
    """The Cout = 3 instantiations of the table query kernels (pifu_query_tabws_kernel<3>,
    pifu_query16_tab_kernel<3, 3>): a 3-channel tanh head on a 256-channel map, table path against the
    plain kernel of the same precision and against the fp32 oracle, ragged sizes included."""
    layers = syn.rand_mlp("G", 34, 2.0)
    rs = np.random.RandomState(5)
    w4 = layers[-1][0]
    layers[-1] = (rs.uniform(-0.1, 0.1, (3, w4.shape[1])).astype(np.float32), rs.uniform(-0.1, 0.1, (3,)).astype(np.float32))
    f = syn.rand_feat(256, 64, 64, 6)
    p = syn.rand_points(5000, 9, 1.1)
    calib = oracle.pifu_calib(*syn.scene_camera(40))
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, layers, 2)  # tanh
    mlp.set_precision(precision)
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    pts, cal = torch.from_numpy(p)[None].to(dev), torch.from_numpy(calib).to(dev)
    plain = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)
    try:
        table = ops.skip_table(mlp, fh)
        out = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)
        for n in (1, 31, 33, 97, 1000):
            part = ops.query(mlp, fh, pts[:, :, :n].contiguous(), cal, syn.Z_SCALE)
            assert torch.equal(part, out[:, :, :n]), n
    finally:
        ops.skip_table_release(mlp.ctx)
    ref = oracle.query(f, p, calib[0], layers, 2, syn.Z_SCALE, precision="f32")
    assert out.shape == (1, 3, 5000) and not torch.equal(out, plain)
    d_plain = (out - plain).abs().max().item()
    d_ref = np.abs(out[0].cpu().numpy() - ref).max()
    print("Cout = 3 %s: |table - plain| %.3g, |table - f32 oracle| %.3g" % (precision, d_plain, d_ref))
    assert d_plain <= 2e-6 and d_ref <= TOL_ORACLE
    del table
