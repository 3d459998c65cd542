"""The C-ABI entry points added in round 6, through ctypes on an MI355X: mp_recon_batch_early (the early hand-over of
the drop-in engine), CU-masked streams, mp_memory_stats, mp_max_frames."""
import ctypes

import numpy as np
import pytest

from monoport_amd import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"
BMIN, BMAX = [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]
RES = [17, 33, 65]


@pytest.fixture(scope="module")
def scene(oracle):
    from monoport_amd import ops
    layers = syn.body_mlp("G", noise=0.05, seed=1)
    mlp = ops.PackedMLP.from_layers(DEV, layers, 1)
    empty = ops.PackedMLP.from_layers(DEV, syn.body_mlp("G", c=-3.0), 1)  # occupancy < 0.5 everywhere
    fh = ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2))[None].to(DEV))
    cal = torch.from_numpy(oracle.pifu_calib(*syn.scene_camera(30))).to(DEV)
    return dict(ops=ops, mlp=mlp, empty=empty, fh=fh, cal=cal)


def test_recon_batch_early_hands_over_after_the_coarsest_level(scene):
    """mp_recon_batch_early: results are mp_recon_batch_ex's; after the coarsest level the host gets (non-empty,
    differs-from-expected) per frame through pinned memory and an event -- `differs` is 0 for the kernel's own
    coarsest-level values, 1 when ONE expected value is off by one ulp, and stays 0 without an expectation; an empty
    scene reports non-empty = 0 (the reference returns None there, RTL/recon.py:32-33)."""
    ops = scene["ops"]
    mlp, fh, cal = scene["mlp"], scene["fh"], scene["cal"]
    vol_ref, st_ref = ops.recon(mlp, fh, cal, syn.Z_SCALE, BMIN, BMAX, RES)
    s = (RES[-1] - 1) // (RES[0] - 1)
    level0 = vol_ref[::s, ::s, ::s].contiguous()  # the coarsest lattice keeps its exact values through the upsampling
    early = ops.EarlyFlags(DEV, 1)
    vol, st = ops.recon(mlp, fh, cal, syn.Z_SCALE, BMIN, BMAX, RES, early=early, expect_level0=level0)
    flags = early.wait().clone()
    assert flags.tolist() == [[1, 0]]
    assert torch.equal(vol, vol_ref) and torch.equal(st, st_ref)
    off = level0.clone()
    off.view(-1)[1234] = torch.nextafter(off.view(-1)[1234], torch.tensor(2.0, device=DEV))
    vol2, _ = ops.recon(mlp, fh, cal, syn.Z_SCALE, BMIN, BMAX, RES, early=early, expect_level0=off)
    assert early.wait().tolist() == [[1, 1]] and torch.equal(vol2, vol_ref)
    ops.recon(mlp, fh, cal, syn.Z_SCALE, BMIN, BMAX, RES, early=early)  # no expectation: only the non-empty flag
    assert early.wait().tolist() == [[1, 0]]
    ops.recon(scene["empty"], fh, cal, syn.Z_SCALE, BMIN, BMAX, RES, early=early)
    assert early.wait()[0, 0].item() == 0
    # a batch: frame 1 is the empty head's scene is not expressible in one call (one head per call) -- two frames of
    # the same head with different cameras, one expectation wrong
    cal2 = scene["cal"].clone()
    early2 = ops.EarlyFlags(DEV, 2)
    vols, sts = ops.recon_batch(mlp, [fh, fh], [cal, cal2], syn.Z_SCALE, BMIN, BMAX, RES, early=early2,
                                expect_level0=[level0, off])
    assert early2.wait().tolist() == [[1, 0], [1, 1]]
    assert torch.equal(vols[0], vol_ref) and torch.equal(vols[1], vol_ref) and torch.equal(sts[0], st_ref)
    with pytest.raises(ValueError):
        ops.recon_batch(mlp, [fh, fh], [cal, cal2], syn.Z_SCALE, BMIN, BMAX, RES, early=early)  # flags for 1 frame


def test_cu_masked_stream_runs_the_same_bits(scene):
    """mp_stream_create_cu_mask: a stream restricted to 64 of the CUs -- the persistent query kernels size their grids
    from its share (mp_stream_cu_count) and give the same bits as on an ordinary stream; bad ranges are refused;
    mp_stream_destroy forgets the stream."""
    from monoport_amd._lib import MonoportError
    ops = scene["ops"]
    mlp, fh, cal = scene["mlp"], scene["fh"], scene["cal"]
    ctx, lib = mlp.ctx, mlp.ctx.lib
    n_cu = lib.mp_stream_cu_count(ctx.handle, None)
    assert n_cu == torch.cuda.get_device_properties(0).multi_processor_count
    vol_ref, st_ref = ops.recon(mlp, fh, cal, syn.Z_SCALE, BMIN, BMAX, RES)
    pts = torch.from_numpy(syn.rand_points(20000, 5, 1.1))[None].to(DEV)
    out_ref = ops.query(mlp, fh, pts, cal, syn.Z_SCALE)
    table = ops.skip_table(mlp, fh)
    vol_tab, _ = ops.recon(mlp, fh, cal, syn.Z_SCALE, BMIN, BMAX, RES)
    h = ctypes.c_void_p()
    ctx.check(lib.mp_stream_create_cu_mask(ctx.handle, 64, 64, ctypes.byref(h)), "mp_stream_create_cu_mask")
    try:
        assert lib.mp_stream_cu_count(ctx.handle, h) == 64
        st = torch.cuda.ExternalStream(h.value, device=DEV)
        st.wait_stream(torch.cuda.current_stream())
        # torch's allocator notes every stream a tensor was used on (record_stream, ops.recon_batch does it for the
        # calibration) and records an event on THAT stream when the tensor is freed: nothing that met the masked stream
        # may outlive it -- private copies of the inputs, gone before the stream is
        cal_m, pts_m = cal.clone(), pts.clone()
        with torch.cuda.stream(st):
            vol_m, _ = ops.recon(mlp, fh, cal_m, syn.Z_SCALE, BMIN, BMAX, RES)       # table kernel, 128 workgroups
            table.release()
            vol_p, st_p = ops.recon(mlp, fh, cal_m, syn.Z_SCALE, BMIN, BMAX, RES)    # plain kernels
            out_p = ops.query(mlp, fh, pts_m, cal_m, syn.Z_SCALE)
            st.synchronize()
        same = (torch.equal(vol_m, vol_tab), torch.equal(vol_p, vol_ref), torch.equal(st_p, st_ref), torch.equal(out_p, out_ref))
        del vol_m, vol_p, st_p, out_p, cal_m, pts_m
        assert all(same), same
    finally:
        table.release()
        torch.cuda.synchronize()
    # The stream torch worked on is NOT destroyed here: torch's allocator keeps per-stream state (cached blocks, the
    # events of record_stream'ed tensors) and would touch a dead stream later -- a caller who wants it gone must drop
    # every tensor that met it and torch.cuda.empty_cache() first (header).  mp_stream_destroy is exercised on a
    # stream only the library itself has used:
    h2 = ctypes.c_void_p()
    ctx.check(lib.mp_stream_create_cu_mask(ctx.handle, 0, 32, ctypes.byref(h2)), "mp_stream_create_cu_mask")
    out4 = (ctypes.c_double * 4)()
    ctx.check(lib.mp_mfma_clock_probe(ctx.handle, ctypes.c_float(5.0), out4, h2), "mp_mfma_clock_probe")
    full = ops.mfma_clock_probe(DEV, 5.0)
    print("clock probe on 32 of %d CUs: %.1f TFLOP/s (whole device %.1f)" % (n_cu, out4[0], full["tflops"]))
    assert int(out4[3]) == 64 and 0.09 < out4[0] / full["tflops"] < 0.16  # 2 workgroups per masked CU; 1/8 of the rate
    ctx.check(lib.mp_stream_destroy(ctx.handle, h2), "mp_stream_destroy")
    assert lib.mp_stream_cu_count(ctx.handle, h2) == n_cu  # forgotten: sized like any other stream
    h = h2
    for first, n in ((0, 4), (-1, 64), (n_cu - 32, 64)):
        with pytest.raises(MonoportError):
            ctx.check(lib.mp_stream_create_cu_mask(ctx.handle, first, n, ctypes.byref(h)), "mp_stream_create_cu_mask")
    with pytest.raises(MonoportError):
        ctx.check(lib.mp_stream_destroy(ctx.handle, ctypes.c_void_p(12345)), "mp_stream_destroy")


def test_memory_stats_and_max_frames(scene):
    """mp_memory_stats counts what the contexts own (a new head adds exactly its packed weights, a registered table one
    entry, a larger reconstruction grows the stream's arena and never shrinks it); mp_max_frames is the batch limit the
    Python side reads."""
    ops = scene["ops"]
    assert ops.MAX_FRAMES == ops.get_context(DEV).lib.mp_max_frames() == 32
    before = ops.memory_stats(DEV)
    extra = ops.PackedMLP.from_layers(DEV, syn.rand_mlp("G", 3, 1.0), 1)
    mid = ops.memory_stats(DEV)
    n_weights = sum(w.size + b.size for w, b in syn.rand_mlp("G", 3, 1.0))
    assert mid["weight_bytes"] - before["weight_bytes"] >= 2 * 4 * (n_weights - 2000)  # packed + raw copies, f32
    table = ops.skip_table(scene["mlp"], scene["fh"])
    assert ops.memory_stats(DEV)["skip_tables"] == mid["skip_tables"] + 1
    table.release()
    assert ops.memory_stats(DEV)["skip_tables"] == mid["skip_tables"]
    s = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(s):
        ops.recon(scene["mlp"], scene["fh"], scene["cal"], syn.Z_SCALE, BMIN, BMAX, [17, 33, 65, 129])
        s.synchronize()
    grown = ops.memory_stats(DEV)  # torch hands streams out of a pool: `s` may have met the library in an earlier test
    assert grown["arenas"] >= mid["arenas"] and grown["arena_bytes"] >= mid["arena_bytes"] and grown["arena_bytes"] > 0
    ops.stream_release(s)
    after = ops.memory_stats(DEV)
    assert after["arenas"] == grown["arenas"] - 1 and after["arena_bytes"] < grown["arena_bytes"]
    del extra
