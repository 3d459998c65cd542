"""Out-of-bounds write check: every output buffer of the C-ABI calls of one reconstruction is
carved out of a poisoned slab with guard bands on both sides; the bands must stay untouched."""
import ctypes

import numpy as np
import pytest

from monoport_amd import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"
GUARD = 4096  # bytes on each side
POISON = 0x5A


class Slab:
    def __init__(self, nbytes):
        self.buf = torch.full((nbytes,), POISON, dtype=torch.uint8, device=DEV)
        self.off = 0
        self.regions = []

    def take(self, nbytes, name):
        start = self.off + GUARD
        start = (start + 255) & ~255
        self.regions.append((name, start, nbytes))
        self.off = start + nbytes
        assert self.off + GUARD <= self.buf.numel()
        return self.buf.data_ptr() + start

    def check(self):
        host = self.buf.cpu().numpy()
        mask = np.ones(host.shape, bool)
        for _, s, n in self.regions:
            mask[s:s + n] = False
        bad = np.nonzero(mask & (host != POISON))[0]
        if bad.size:
            first = int(bad[0])
            near = [(n, s, s + k) for n, s, k in self.regions if s - 2 * GUARD <= first <= s + k + 2 * GUARD]
            raise AssertionError("guard band overwritten at byte %d (%d bytes), near %s" % (first, bad.size, near))


@pytest.mark.parametrize("res", [[9, 17, 33], [17, 33, 65, 129]])
def test_no_out_of_bounds_writes(res):
    from monoport_amd import ops
    ctx = ops.get_context(DEV)
    lib, h = ctx.lib, ctx.handle
    r = res[-1]
    mlp = ops.PackedMLP.from_layers(DEV, syn.body_mlp("G", noise=0.05, seed=1), 1)
    mlpc = ops.PackedMLP.from_layers(DEV, syn.rand_mlp("C", 3, 1.0), 2)
    feat = torch.from_numpy(syn.body_feat(256, 128, 128, 2))[None].to(DEV)
    featc = torch.from_numpy(syn.rand_feat(512, 128, 128, 4))[None].to(DEV)
    from oracle import pifu_oracle as orc
    calib = torch.from_numpy(orc.pifu_calib(*syn.scene_camera(0))).to(DEV)[0].contiguous()
    slab = Slab(64 << 20 if r <= 33 else 160 << 20)
    cap = r * r
    p_hwc = slab.take(128 * 128 * 256 * 4, "feat_hwc")
    p_hwcc = slab.take(128 * 128 * 512 * 4, "feat_hwc_c")
    p_vol = slab.take(r ** 3 * 4, "volume")
    p_status = slab.take(4 * (1 + len(res)), "status")
    p_x = slab.take(cap * 8, "X")
    p_y = slab.take(cap * 8, "Y")
    p_z = slab.take(cap * 4, "Z")
    p_n = slab.take(cap * 12, "norm")
    p_cnt = slab.take(4, "count")
    p_img = slab.take(r * r * 12, "image")
    p_pts = slab.take(cap * 12, "points")
    p_pred = slab.take(cap * 12, "preds")
    p_img2 = slab.take(r * r * 12, "image2")
    n_q = 1000
    p_qout = slab.take(n_q * 4, "query_out")
    vp = ctypes.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    ctx.check(lib.mp_feat_pack_hwc(h, vp(feat.data_ptr()), 256, 128, 128, vp(p_hwc), 256, 0, st), "pack")
    ctx.check(lib.mp_feat_pack_hwc(h, vp(featc.data_ptr()), 512, 128, 128, vp(p_hwcc), 512, 0, st), "pack")
    pts = torch.from_numpy(syn.rand_points(n_q, 9, 1.1)).to(DEV)
    ctx.check(lib.mp_query(h, mlp.id, vp(p_hwc), 256, 128, 128, vp(pts.data_ptr()), n_q, 1, n_q,
                           vp(calib.data_ptr()), 1.28, vp(p_qout), st), "query")
    bmin = (ctypes.c_float * 3)(-1, -1, -1)
    bmax = (ctypes.c_float * 3)(1, 1, 1)
    resc = (ctypes.c_int * len(res))(*res)
    ctx.check(lib.mp_recon(h, mlp.id, vp(p_hwc), 256, 128, 128, vp(calib.data_ptr()), 1.28, bmin, bmax,
                           resc, len(res), 0.5, vp(p_vol), vp(p_status), st), "recon")
    ctx.check(lib.mp_forward_vertices(h, vp(p_vol), r, 0, vp(p_x), vp(p_y), vp(p_z), vp(p_n), vp(p_cnt), st), "fv")
    ctx.check(lib.mp_paint(h, vp(p_x), vp(p_y), vp(p_n), 0, vp(p_cnt), cap, r, 0.5, 0.5, 0.0, 1.0, vp(p_img), st), "paint")
    mat = np.eye(4, dtype=np.float32)
    mat[0, 0] = mat[1, 1] = mat[2, 2] = 2.0 / r
    mat[:3, 3] = -1
    matc = (ctypes.c_float * 16)(*mat.reshape(-1))
    ctx.check(lib.mp_vertex_points(h, vp(p_x), vp(p_y), vp(p_z), vp(p_cnt), cap, r, matc, vp(p_pts), st), "vp")
    ctx.check(lib.mp_query_counted(h, mlpc.id, vp(p_hwcc), 512, 128, 128, vp(p_pts), cap, vp(p_cnt),
                                   vp(calib.data_ptr()), 1.28, vp(p_pred), st), "qc")
    ctx.check(lib.mp_paint(h, vp(p_x), vp(p_y), vp(p_pred), 1, vp(p_cnt), cap, r, 0.5, 0.5, -1e30, 1e30, vp(p_img2), st), "paint2")
    torch.cuda.synchronize()
    slab.check()


def test_mfma_clock_probe_reports_a_plausible_rate_and_clock():
    """mp_mfma_clock_probe (bench.py's `roofline.sustained`): the register-only v_mfma_f32_32x32x2_f32 loop must land
    near the nominal 157.3 TFLOP/s at a shader clock near 2.4 GHz -- and rate / clock must be the 256 CUs x 4 SIMDs x
    64 FLOP per cycle of the matrix pipe (the probe measures both independently: events vs in-kernel counters)."""
    from monoport_amd import ops
    from monoport_amd._lib import MonoportError
    got = ops.mfma_clock_probe(DEV, 10.0)
    print("mfma clock probe:", got)
    assert 5.0 < got["ms"] < 40.0 and got["workgroups"] == 2 * torch.cuda.get_device_properties(0).multi_processor_count
    assert 1500.0 < got["shader_clock_mhz"] < 2600.0 and 100.0 < got["tflops"] < 165.0
    per_clock = got["tflops"] * 1e12 / (got["shader_clock_mhz"] * 1e6) / (got["workgroups"] // 2 * 4)
    assert 60.0 < per_clock < 65.0, per_clock  # FLOP per SIMD and cycle: 64 when the pipe never idles
    with pytest.raises(MonoportError):
        ops.mfma_clock_probe(DEV, 0.0)
