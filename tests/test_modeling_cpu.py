"""Host-side mirror of the reference's module API (CPU only: construction, state-dict layout,
encoder numerics against reference-generated goldens, argument validation)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from monoport_amd import synthetic as syn


def test_netg_netc_structure():
    from monoport_amd.modeling import PIFuNetC, PIFuNetG
    g, c = PIFuNetG().eval(), PIFuNetC().eval()
    # parameter counts of the reference modules
    assert sum(p.numel() for p in g.surface_classifier.parameters()) == 1183874
    assert sum(p.numel() for p in c.surface_classifier.parameters()) == 1676934
    shapes_g = [tuple(f.weight.shape) for f in g.surface_classifier.filters]
    shapes_c = [tuple(f.weight.shape) for f in c.surface_classifier.filters]
    assert shapes_g == [(1024, 257, 1), (512, 1281, 1), (256, 769, 1), (128, 513, 1), (1, 385, 1)]
    assert shapes_c == [(1024, 513, 1), (512, 1537, 1), (256, 1025, 1), (128, 769, 1), (3, 641, 1)]
    assert len(g.image_filter.state_dict()) == 431 and len(c.image_filter.state_dict()) == 43
    for attr in ("image_filter", "surface_classifier", "projection", "normalizer"):
        assert hasattr(g, attr)
    assert abs(g.normalizer.scale - 1.28) < 1e-12


def test_encoders_match_reference_goldens():
    """netG.filter / netC.filter reproduce the reference's outputs under the same seeded weights
    (fixture made by oracle/gen_golden.py running the reference modules)."""
    from monoport_amd.modeling import PIFuNetC, PIFuNetG
    g = load_golden("encoders")
    netg, netc = PIFuNetG().eval(), PIFuNetC().eval()
    for net, seed in ((netg, 71), (netc, 72)):
        shapes = {k: tuple(v.shape) for k, v in net.image_filter.state_dict().items()}
        sd = syn.seeded_state_dict(shapes, seed)
        net.image_filter.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    img = torch.from_numpy(syn.synthetic_image(73))[None]
    with torch.no_grad():
        fg = netg.filter(img)
        fc = netc.filter(img, feat_prior=fg[-1][-1])
    assert len(fg) == 4 and len(fc) == 1 and fc[0][0].shape == (1, 512, 128, 128)
    for i in range(4):
        assert np.abs(fg[i][0][0, ::8, ::8, ::8].numpy() - g["G%d" % i]).max() <= 1e-4
    assert np.abs(fc[0][0][0, ::8, ::8, ::8].numpy() - g["C0"]).max() <= 1e-4
    from conftest import check_full_coverage
    check_full_coverage(g, "G3", fg[-1][0][0].numpy())
    check_full_coverage(g, "C0", fc[0][0][0].numpy())
    # prior first (MonoPortNet.py:44)
    assert torch.equal(fc[0][0][:, :256], fg[-1][-1])


def test_legacy_checkpoint_key_mapping(tmp_path):
    from monoport_amd.modeling import PIFuNetG
    net = PIFuNetG().eval()
    ckpt = {}
    for k, v in net.image_filter.state_dict().items():
        ckpt["image_filter." + k] = torch.full_like(v, 0.5)
    for k, v in net.surface_classifier.state_dict().items():
        ckpt["surface_classifier." + k.replace("filters.", "conv")] = torch.full_like(v, 0.25)
    path = tmp_path / "legacy.pt"
    torch.save(ckpt, path)
    net.load_legacy_pifu(str(path))
    assert float(net.surface_classifier.filters[2].weight.detach().mean()) == 0.25
    assert float(net.image_filter.conv1.weight.detach().mean()) == 0.5


def test_pifu_calib_cpu():
    from monoport_amd.recon import pifu_calib
    g = load_golden("pifu_calib")
    for step, ref in zip(g["steps"], g["calib"]):
        ext, intr = syn.scene_camera(int(step))
        e0, i0 = ext.copy(), intr.copy()
        out = pifu_calib(ext, intr, device="cpu")
        assert out.shape == (1, 4, 4) and out.dtype == torch.float32
        assert np.array_equal(out[0].numpy(), ref)
        assert np.array_equal(e0, ext) and np.array_equal(i0, intr)


def test_forward_vertices_and_colorization_none_passthrough():
    from monoport_amd.recon import colorization, forward_vertices
    assert forward_vertices(None) == (None, None, None, None)
    assert colorization(None, None, None, None, None, None) is None


def test_seg3d_ctor_validation():
    from monoport_amd.implicit_seg.functional import Seg3dLossless, Seg3dTopk
    bmin, bmax = np.array([[-1., -1., -1.]]), np.array([[1., 1., 1.]])
    eng = Seg3dLossless(query_func=lambda **kw: None, b_min=bmin, b_max=bmax,
                        resolutions=[17, 33, 65, 129, 257], balance_value=0.5,
                        use_cuda_impl=False, faster=True)
    assert isinstance(eng, torch.nn.Module) and eng.resolutions[-1] == 257
    eng.to("cpu")
    with pytest.raises(AssertionError):
        Seg3dLossless(lambda **kw: None, bmin, bmax, [16, 31])
    with pytest.raises(NotImplementedError):
        Seg3dLossless(lambda **kw: None, bmin, bmax, [17, 35])
    with pytest.raises(NotImplementedError):
        Seg3dTopk()


def test_implicit_seg_aliasing():
    import sys
    import monoport_amd.implicit_seg as iseg
    sys.modules.setdefault("implicit_seg", iseg)
    sys.modules.setdefault("implicit_seg.functional", iseg.functional)
    sys.modules.setdefault("implicit_seg.functional.utils", iseg.functional.utils)
    from implicit_seg.functional import Seg3dLossless, Seg3dTopk  # noqa: F401
    from implicit_seg.functional.utils import plot_mask3D  # noqa: F401


def test_query_on_cpu_fails_loudly():
    from monoport_amd._lib import MonoportError
    from monoport_amd.modeling import PIFuNetG
    net = PIFuNetG().eval()
    feats = [[torch.zeros(1, 256, 128, 128)]]
    with pytest.raises(MonoportError):
        net.query(feats, torch.zeros(1, 3, 8), torch.eye(4)[None])


def test_skip_table_handle_supersession_cpu():
    """ops.SkipTable (host logic only, fake context): a handle unregisters a map only while it is the
    LAST registration of that map -- re-making the table of the same buffers (next frame) must not be
    undone when the older handle is dropped -- and release is idempotent."""
    import ctypes
    import torch
    from monoport_amd import ops

    calls = []

    class FakeLib:
        def mp_skip_table_release(self, handle, feat, table):
            calls.append((feat.value, table.value))
            return 0

    class FakeCtx:
        lib = FakeLib()
        handle = ctypes.c_void_p(1234)

    ctx = FakeCtx()
    feat = torch.zeros((4, 4, 256))
    table = torch.zeros((4, 4, ops.SKIP_TABLE_ROWS))
    h1 = ops.SkipTable(ctx, [feat], table)
    h2 = ops.SkipTable(ctx, [feat], table)  # same buffers, next frame
    h1.release()
    assert calls == []  # superseded: nothing to undo
    h1.release()
    h2.release()
    assert calls == [(feat.data_ptr(), table.data_ptr())]
    h2.release()
    assert len(calls) == 1
    # a batch handle: one registration per map, table views per map
    feats = torch.zeros((3, 4, 4, 256))
    tables = torch.zeros((3, 4, 4, ops.SKIP_TABLE_ROWS))
    hb = ops.SkipTable(ctx, [feats[i] for i in range(3)], tables)
    del hb  # collection releases
    assert [c[0] for c in calls[1:]] == [feats[i].data_ptr() for i in range(3)]
    assert [c[1] for c in calls[1:]] == [tables[i].data_ptr() for i in range(3)]
