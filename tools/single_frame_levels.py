"""Per octree level, the fused-query launches of ONE frame (what the reference's per-frame recon stage issues,
RTL/main.py:389-395) against the same levels at 20 frames per launch: points, ms, fraction of the f32 MFMA roof on
the executed FLOPs of the skip-table kernel.

  python tools/single_frame_levels.py [frames ...]        (on the GPU box; default 1 2 4 20)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from monoport_amd import ops, synthetic as syn
from monoport_amd.recon import pifu_calib

dev = torch.device("cuda", 0)
PEAK = 157.3e12
FPP = 1380354  # executed FLOP per point of pifu_query_tabws_kernel (bench.py: roofline.flops_per_point_executed)
netG, _ = bench.build_netg(dev)
mlp = netG.surface_classifier.packed()
RES = bench.RESOLUTIONS
frames = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 20]
for nf in frames:
    fmaps = torch.stack([torch.from_numpy(syn.body_feat(256, 128, 128, 2 + i)).permute(1, 2, 0).contiguous() for i in range(nf)]).to(dev)
    cals = [pifu_calib(*syn.scene_camera(30 + 3 * i), device=dev) for i in range(nf)]
    tab = ops.skip_table_batch(mlp, fmaps)
    feats = [fmaps[i] for i in range(nf)]
    for _ in range(3):
        vols, st = ops.recon_batch(mlp, feats, cals, syn.Z_SCALE, bench.B_MIN, bench.B_MAX, RES)
    torch.cuda.synchronize()
    reps = 10
    ops.profile_begin(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        vols, st = ops.recon_batch(mlp, feats, cals, syn.Z_SCALE, bench.B_MIN, bench.B_MAX, RES, volumes=vols, status=st)
    ev1.record()
    torch.cuda.synchronize()
    ms = np.array(ops.profile_end(dev)).reshape(reps, len(RES))
    pts = st.cpu().numpy()[:, 1:].sum(0).astype(np.float64)
    print("%2d frame(s) per launch: whole mp_recon_batch %.3f ms per frame (%.3f ms per call); query launches %.3f ms per frame"
          % (nf, ev0.elapsed_time(ev1) / reps / nf, ev0.elapsed_time(ev1) / reps, ms.sum(1).mean() / nf))
    for l in range(len(RES)):
        m = ms[:, l].mean()
        print("    level %d: %8.0f points, %3d tiles of 32, %7.3f ms, frac %.3f" % (l, pts[l], int(np.ceil(pts[l] / 32)), m, pts[l] * FPP / (m * 1e-3) / PEAK))
    print("    all levels: frac %.3f" % (pts.sum() * FPP / (ms.sum(1).mean() * 1e-3) / PEAK))
    tab.release()
