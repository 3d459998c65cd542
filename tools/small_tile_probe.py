"""The fused f32 query on 64-point tiles (query.hip, mp_query_tune(0)) against 32-point tiles
(query_small.hip, mp_query_tune(1)): mp_query on N points over a sweep of N, and the five octree levels
of 1 / 2 / 4 / 16 frames through mp_recon(_batch).

    python tools/small_tile_probe.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monoport_amd import _lib, ops, synthetic as syn  # noqa: E402
from monoport_amd.recon import pifu_calib  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
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


def main():
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    mlp = ops.PackedMLP.from_layers(dev, syn.body_mlp("G", noise=0.05, seed=1), 1)
    fh = ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2))[None].to(dev))
    cal = pifu_calib(*syn.scene_camera(30), device=dev)
    print("points  tiles64   64-pt ms   32-pt ms   ratio")
    for tiles in (16, 77, 142, 200, 256, 300, 392, 450, 512, 600, 768, 1024, 1536, 2048, 4096):
        n = tiles * 64
        p = torch.from_numpy(syn.rand_points(n, 7, 1.0))[None].to(dev)
        t = {}
        for mode in (0, 1):
            lib.mp_query_tune(mode)
            t[mode] = timed(lambda: ops.query(mlp, fh, p, cal, syn.Z_SCALE))
        lib.mp_query_tune(-1)
        print("%7d %7d %10.3f %10.3f %7.2f" % (n, tiles, t[0], t[1], t[1] / t[0]))
    res = [17, 33, 65, 129, 257]
    for frames in (1, 2, 4, 16):
        t = {}
        for mode in (0, 512, 1024, 4096, 1):
            lib.mp_query_tune(mode)
            if frames == 1:
                t[mode] = timed(lambda: ops.recon(mlp, fh, cal, syn.Z_SCALE, [-1] * 3, [1] * 3, res))
            else:
                t[mode] = timed(lambda: ops.recon_batch(mlp, [fh] * frames, [cal] * frames, syn.Z_SCALE,
                                                        [-1] * 3, [1] * 3, res), reps=10)
        lib.mp_query_tune(-1)
        print("mp_recon x%d frames, 257^3: 64-point tiles %.3f ms  gate 512 / 1024 / 4096: %.3f / %.3f / %.3f ms"
              "   32-point tiles %.3f ms" % (frames, t[0], t[512], t[1024], t[4096], t[1]))


if __name__ == "__main__":
    main()
