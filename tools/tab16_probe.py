"""Round 4: the split-precision query kernels through the skip table (pifu_query16_tab_kernel) against their
plain twins (pifu_query16_kernel): one launch of 885 k lattice points and mp_recon_batch of 16 frames at
257^3 per precision, with the volumes' differences against the exact-f32 table path.

    python tools/tab16_probe.py [quick]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monoport_amd import ops, synthetic as syn  # noqa: E402
from monoport_amd.recon import pifu_calib  # noqa: E402
from skip_table_probe import lattice_points, timed  # noqa: E402

FLOP = 2363906


def main():
    os.environ["MONOPORT_TAB16"] = "all"  # by default only f16x3 is routed through the tables
    quick = "quick" in sys.argv[1:]
    dev = torch.device("cuda", 0)
    mlp = ops.PackedMLP.from_layers(dev, syn.body_mlp("G", noise=0.05, seed=1), 1)
    frames = 2 if quick else 16
    feats = [ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2 + i))[None].to(dev)) for i in range(frames)]
    cal = pifu_calib(*syn.scene_camera(30), device=dev)
    p = torch.from_numpy(lattice_points(32 if quick else 96))[None].to(dev)
    n = p.shape[2]
    res = [17, 33, 65] if quick else [17, 33, 65, 129, 257]
    tables = torch.empty((frames, 128, 128, ops.SKIP_TABLE_ROWS), device=dev)

    def run():
        return (ops.query(mlp, feats[0], p, cal, syn.Z_SCALE),
                ops.recon_batch(mlp, feats, [cal] * frames, syn.Z_SCALE, [-1] * 3, [1] * 3, res)[0])

    handles = [ops.skip_table(mlp, feats[i], out=tables[i]) for i in range(frames)]
    q32, v32 = run()
    q32, v32 = q32.clone(), [v.clone() for v in v32]
    ops.skip_table_release(mlp.ctx)
    print("%d lattice points per launch; mp_recon_batch of %d frames" % (n, frames))
    for prec in ("f16x3", "f16w", "f16"):
        mlp.set_precision(prec)
        row = {}
        for path in ("plain", "table"):
            if path == "table":
                handles = [ops.skip_table(mlp, feats[i], out=tables[i]) for i in range(frames)]
            q, v = run()
            torch.cuda.synchronize()
            dq = (q - q32).abs().max().item()
            dv = max((a - b).abs().max().item() for a, b in zip(v, v32))
            flips = sum(int(((a > 0.5) != (b > 0.5)).sum()) for a, b in zip(v, v32))
            if quick:
                row[path] = (0.0, 0.0, dq, dv, flips)
            else:
                tq = timed(lambda: ops.query(mlp, feats[0], p, cal, syn.Z_SCALE))
                tr = timed(lambda: ops.recon_batch(mlp, feats, [cal] * frames, syn.Z_SCALE, [-1] * 3, [1] * 3, res), reps=5)
                row[path] = (tq, tr, dq, dv, flips)
            if path == "table":
                ops.skip_table_release(mlp.ctx)
        for path, (tq, tr, dq, dv, flips) in row.items():
            print("  %-5s %-5s %8.3f ms per launch = %6.1f TFLOP/s-equivalent | recon_batch %8.3f ms = %.3f ms per frame | "
                  "vs f32 table path: query %.3g, volumes %.3g, %d voxels on the other side of 0.5"
                  % (prec, path, tq, n * FLOP / tq / 1e9 if tq else 0.0, tr, tr / frames, dq, dv, flips), flush=True)
    mlp.set_precision("f32")
    del handles


if __name__ == "__main__":
    main()
