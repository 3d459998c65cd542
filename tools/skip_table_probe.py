"""The skip-table path (mp_skip_table + pifu_query_tab_kernel, csrc/query_table.hip) against the
plain fused kernels: a fixed set of lattice points, then mp_recon_batch of 16 frames.

    python tools/skip_table_probe.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monoport_amd import _lib  # noqa: E402
from monoport_amd import ops, synthetic as syn  # noqa: E402
from monoport_amd.recon import pifu_calib  # noqa: E402


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def lattice_points(n=96):
    """n^3 lattice points at the finest level's spacing (2/256), in 4 x 4 x 4 blocks."""
    b = np.arange(n // 4)
    bz, by, bx = np.meshgrid(b, b, b, indexing="ij")
    l = np.arange(4)
    lz, ly, lx = np.meshgrid(l, l, l, indexing="ij")
    z = (bz.reshape(-1, 1) * 4 + lz.reshape(1, -1)).reshape(-1)
    y = (by.reshape(-1, 1) * 4 + ly.reshape(1, -1)).reshape(-1)
    x = (bx.reshape(-1, 1) * 4 + lx.reshape(1, -1)).reshape(-1)
    return (np.stack([x, y, z]).astype(np.float32) - n / 2) * (2.0 / 256)


def main():
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    mlp = ops.PackedMLP.from_layers(dev, syn.body_mlp("G", noise=0.05, seed=1), 1)
    frames = 16
    feats = [ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2 + i))[None].to(dev)) for i in range(frames)]
    cal = pifu_calib(*syn.scene_camera(30), device=dev)
    pts = lattice_points()
    p = torch.from_numpy(pts)[None].to(dev)
    n = pts.shape[1]
    res = [17, 33, 65, 129, 257]
    rows = {}
    for gate, name in ((0, "64-point tiles"), (1, "32-point tiles")):
        lib.mp_query_tune(gate)
        rows[name] = (timed(lambda: ops.query(mlp, feats[0], p, cal, syn.Z_SCALE)),
                      timed(lambda: ops.recon_batch(mlp, feats, [cal] * frames, syn.Z_SCALE, [-1] * 3, [1] * 3, res), reps=5))
    lib.mp_query_tune(-1)
    rows["default gate"] = (timed(lambda: ops.query(mlp, feats[0], p, cal, syn.Z_SCALE)),
                            timed(lambda: ops.recon_batch(mlp, feats, [cal] * frames, syn.Z_SCALE, [-1] * 3, [1] * 3, res), reps=5))
    tables = torch.empty((frames, 128, 128, ops.SKIP_TABLE_ROWS), device=dev)
    handles = [None] * frames  # the registrations live as long as these handles do

    def make_tables():
        for i in range(frames):
            handles[i] = ops.skip_table(mlp, feats[i], out=tables[i])

    t_tab = timed(make_tables)
    rows["skip table"] = (timed(lambda: ops.query(mlp, feats[0], p, cal, syn.Z_SCALE)),
                             timed(lambda: ops.recon_batch(mlp, feats, [cal] * frames, syn.Z_SCALE, [-1] * 3, [1] * 3, res), reps=5))
    ops.skip_table_release(mlp.ctx)
    print("%d lattice points (one launch) / mp_recon_batch of %d frames at 257^3" % (n, frames))
    for name, (tq, tr) in rows.items():
        print("  %-16s %8.3f ms  %6.1f TFLOP/s-equivalent   |  %8.3f ms = %.3f ms per frame"
              % (name, tq, n * 2363906 / tq / 1e9, tr, tr / frames))
    print("  mp_skip_table of %d maps: %.3f ms = %.3f ms per frame (%.1f TFLOP/s)"
          % (frames, t_tab, t_tab / frames, frames * 128 * 128 * 1921 * 256 * 2 / t_tab / 1e9))


if __name__ == "__main__":
    main()
