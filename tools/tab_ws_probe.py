"""Round 4: the wave-specialised skip-table query kernel (pifu_query_tabws_kernel, the default) against
round 3's pifu_query_tab_kernel (MONOPORT_TAB_KERNEL=v1), in one process: agreement of the two fields,
one launch of 885 k lattice points, mp_recon_batch of 16 frames, mp_recon of one frame.

    python tools/tab_ws_probe.py [quick]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monoport_amd import _lib  # noqa: E402
if os.environ.get("MONOPORT_ABLATE"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmp_ablate%s.so" % os.environ["MONOPORT_ABLATE"])
from monoport_amd import ops, synthetic as syn  # noqa: E402
from monoport_amd.recon import pifu_calib  # noqa: E402
from skip_table_probe import lattice_points, timed  # noqa: E402

EXEC_FLOP = 1380354  # executed per point on the table path


def main():
    quick = "quick" in sys.argv[1:]
    dev = torch.device("cuda", 0)
    mlp = ops.PackedMLP.from_layers(dev, syn.body_mlp("G", noise=0.05, seed=1), 1)
    frames = 2 if quick else 16
    feats = [ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2 + i))[None].to(dev)) for i in range(frames)]
    cal = pifu_calib(*syn.scene_camera(30), device=dev)
    pts = lattice_points(32 if quick else 96)
    p = torch.from_numpy(pts)[None].to(dev)
    n = pts.shape[1]
    res = [17, 33, 65] if quick else [17, 33, 65, 129, 257]
    plain = ops.query(mlp, feats[0], p, cal, syn.Z_SCALE).clone()
    tables = torch.empty((frames, 128, 128, ops.SKIP_TABLE_ROWS), device=dev)
    handles = [ops.skip_table(mlp, feats[i], out=tables[i]) for i in range(frames)]
    out, vols, rows = {}, {}, {}
    # a launch shaped like octree level 0 of a 16-frame batch: 78,608 points, every one with its own texels
    sc = torch.from_numpy(syn.rand_points(78608, 3, 0.95))[None].to(dev)
    if os.environ.get("MONOPORT_ABLATE"):  # a side build (possibly with wrong results): timing only
        ts = timed(lambda: ops.query(mlp, feats[0], sc, cal, syn.Z_SCALE), reps=20)
        print("  %-10s scattered 78608 points: %.3f ms = %.3f of 157.3" % (os.environ["MONOPORT_ABLATE"], ts, 78608 * EXEC_FLOP / ts / 1e9 / 157.3))
        tq = timed(lambda: ops.query(mlp, feats[0], p, cal, syn.Z_SCALE))
        tr = timed(lambda: ops.recon_batch(mlp, feats, [cal] * frames, syn.Z_SCALE, [-1] * 3, [1] * 3, res), reps=5)
        print("  %-10s %8.3f ms per %d points = %6.1f TFLOP/s executed (%.3f of 157.3)  |  recon_batch x%d %8.3f ms = %.3f ms per frame"
              % (os.environ["MONOPORT_ABLATE"], tq, n, n * EXEC_FLOP / tq / 1e9, n * EXEC_FLOP / tq / 1e9 / 157.3, frames, tr, tr / frames))
        return
    for name in ("v1", "ws"):
        os.environ["MONOPORT_TAB_KERNEL"] = name
        out[name] = ops.query(mlp, feats[0], p, cal, syn.Z_SCALE).clone()
        # ragged sizes: tails, tiny launches
        for m in (1, 31, 33, 1000, 4913):
            a = ops.query(mlp, feats[0], p[:, :, :m].contiguous(), cal, syn.Z_SCALE)
            assert torch.equal(a, out[name][:, :, :m]), (name, m)
        vols[name] = [v.clone() for v in ops.recon_batch(mlp, feats, [cal] * frames, syn.Z_SCALE, [-1] * 3, [1] * 3, res)[0]]
        torch.cuda.synchronize()
        print("%s: |table - plain| = %.3g" % (name, (out[name] - plain).abs().max().item()), flush=True)
        if quick:
            continue
        rows[name] = (timed(lambda: ops.query(mlp, feats[0], p, cal, syn.Z_SCALE)),
                      timed(lambda: ops.recon_batch(mlp, feats, [cal] * frames, syn.Z_SCALE, [-1] * 3, [1] * 3, res), reps=5),
                      timed(lambda: ops.recon(mlp, feats[0], cal, syn.Z_SCALE, [-1] * 3, [1] * 3, res), reps=5))
    print("|ws - v1| = %.3g on %d points; volumes: max %.3g, voxels on the other side of 0.5: %d"
          % ((out["ws"] - out["v1"]).abs().max().item(), n,
             max((a - b).abs().max().item() for a, b in zip(vols["ws"], vols["v1"])),
             sum(int(((a > 0.5) != (b > 0.5)).sum()) for a, b in zip(vols["ws"], vols["v1"]))))
    for name in ("v1", "ws"):
        os.environ["MONOPORT_TAB_KERNEL"] = name
        ts = timed(lambda: ops.query(mlp, feats[0], sc, cal, syn.Z_SCALE), reps=20)
        print("  %-3s scattered 78608 points: %.3f ms = %.3f of 157.3" % (name, ts, 78608 * EXEC_FLOP / ts / 1e9 / 157.3))
    for name, (tq, tr, t1) in rows.items():
        print("  %-3s %8.3f ms per %d points = %6.1f TFLOP/s executed (%.3f of 157.3)  |  recon_batch x%d %8.3f ms = %.3f ms per frame  |  one frame %.3f ms"
              % (name, tq, n, n * EXEC_FLOP / tq / 1e9, n * EXEC_FLOP / tq / 1e9 / 157.3, frames, tr, tr / frames, t1))
    del handles
    os.environ.pop("MONOPORT_TAB_KERNEL")


if __name__ == "__main__":
    main()
