"""Where the waves of pifu_query_tabws_kernel wait: a side build with -DMPT_WS_STAMP (tools/ablate.py build wsstamp
query_table.hip -DMPT_WS_STAMP) sums, per barrier of the tile schedule, the s_memtime ticks wave 0 (consumer) and
wave 4 (producer) of workgroup 0 WORKED before arriving and WAITED at it, and leaves the sums in the first 64 outputs.

    MONOPORT_ABLATE=wsstamp python tools/tab_ws_stamp_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monoport_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmp_ablate%s.so" % os.environ.get("MONOPORT_ABLATE", "wsstamp"))
from monoport_amd import ops, synthetic as syn  # noqa: E402
from monoport_amd.recon import pifu_calib  # noqa: E402
from skip_table_probe import lattice_points  # noqa: E402

NAMES = ["S0", "S1", "S2", "S3", "S4", "S5", "S6", "S7", "T0", "T1", "T2", "T3", "U0", "U1", "end", "start"]


def show(title, out, ms):
    st = out.flatten()[:64].cpu().double().reshape(2, 16, 2)
    tot = st.sum(dim=(1, 2))
    print("%s: %.3f ms; ticks of workgroup 0: consumer %.0f, producer %.0f" % (title, ms, tot[0], tot[1]))
    print("   barrier   consumer work  wait (%% of its total)   producer work  wait")
    for i, nm in enumerate(NAMES):
        print("   %-6s   %12.0f %6.0f (%4.1f %%)   %12.0f %6.0f (%4.1f %%)"
              % (nm, st[0, i, 0], st[0, i, 1], 100 * st[0, i, 1] / tot[0], st[1, i, 0], st[1, i, 1], 100 * st[1, i, 1] / tot[1]))
    print("   marks inside S7 (producer wave 0, ticks since the barrier, loads drained):", [int(v) for v in out.flatten()[64:72].cpu().tolist()])
    print("   waits: consumer %.1f %%, producer %.1f %%" % (100 * st[0, :, 1].sum() / tot[0], 100 * st[1, :, 1].sum() / tot[1]))


def main():
    dev = torch.device("cuda", 0)
    mlp = ops.PackedMLP.from_layers(dev, syn.body_mlp("G", noise=0.05, seed=1), 1)
    feat = ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2))[None].to(dev))
    cal = pifu_calib(*syn.scene_camera(30), device=dev)
    handle = ops.skip_table(mlp, feat)
    for title, pts in (("885 k lattice points", torch.from_numpy(lattice_points(96))[None].to(dev)),
                       ("78,608 scattered points", torch.from_numpy(syn.rand_points(78608, 3, 0.95))[None].to(dev)),
                       ("110 k lattice points", torch.from_numpy(lattice_points(48))[None].to(dev))):
        for _ in range(3):
            ops.query(mlp, feat, pts, cal, syn.Z_SCALE)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = ops.query(mlp, feat, pts, cal, syn.Z_SCALE)
        e1.record()
        torch.cuda.synchronize()
        show(title, out, e0.elapsed_time(e1))
    del handle
    # the finest level of a 16-frame reconstruction (its launch is the last one that writes into frame 0's volume)
    frames = 16
    feats = [ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2 + i))[None].to(dev)) for i in range(frames)]
    tables = torch.empty((frames, 128, 128, ops.SKIP_TABLE_ROWS), device=dev)
    handles = [ops.skip_table(mlp, feats[i], out=tables[i]) for i in range(frames)]
    res = [17, 33, 65, 129, 257]
    for _ in range(2):
        vols = ops.recon_batch(mlp, feats, [cal] * frames, syn.Z_SCALE, [-1] * 3, [1] * 3, res)[0]
    torch.cuda.synchronize()
    show("level 4 of a 16-frame reconstruction", vols[0], float("nan"))
    del handles


if __name__ == "__main__":
    main()
