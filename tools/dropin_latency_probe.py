"""Where a single frame's latency goes on the reference's call surface (bench.py --mode dropin):
each stage of the RTL/main.py processors list timed with a device synchronize after it.

  python tools/dropin_latency_probe.py     (on the GPU box)

Round 2: H2D 0.09 | calib 0.10 | normalise 0.06 | netG.filter 5.36 | Seg3dLossless 6.31 |
forward_vertices 0.11 | colorization 0.06 ms.  The encoder at batch 1 is GPU-bound, not launch-bound
(all four stacks' outputs, 32^2 / 64^2 maps that fill a fraction of the chip): replaying it as a
hipGraph measured 5.63 ms.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from monoport_amd import synthetic as syn
from monoport_amd.implicit_seg.functional import Seg3dLossless
from monoport_amd.recon import colorization, forward_vertices, pifu_calib

dev = torch.device("cuda", 0)
netG, _ = bench.build_netg(dev)
planes = torch.from_numpy(syn.body_feature_planes(128, 128)).to(dev)


def query_func(points, im_feat_list, calib_tensor):
    samples = points.repeat(1, 1, 1).permute(0, 2, 1)
    return netG.query(im_feat_list, points=samples, calibs=calib_tensor)[0]


engine = Seg3dLossless(query_func=query_func, b_min=np.array([bench.B_MIN], np.float32),
                       b_max=np.array([bench.B_MAX], np.float32), resolutions=bench.RESOLUTIONS,
                       balance_value=0.5, use_cuda_impl=False, faster=True).to(dev)
img = torch.from_numpy(syn.synthetic_image(0))
mask = (img.abs().sum(0, keepdim=True) > 0).float()
frame = torch.cat([img, mask], 0)[None].pin_memory()


def filt(d):
    feats = netG.filter(d["input_netG"])
    feats[-1][0][0, 0:2].copy_(planes)
    return {**d, "feat": feats}


stages = [
    ("H2D", lambda d: {"input": d.to(dev, non_blocking=True)}),
    ("calib", lambda d: {**d, "calib": pifu_calib(*syn.scene_camera(0), device=dev)}),
    ("normalise", lambda d: {**d, "input_netG": (((d["input"][:, 0:3] * 0.5 + 0.5) - 0.5) / 0.5) * d["input"][:, 3:4]}),
    ("netG.filter", filt),
    ("Seg3dLossless", lambda d: {**d, "sdf": engine(im_feat_list=d["feat"], calib_tensor=d["calib"])}),
    ("forward_vertices", lambda d: {**d, **dict(zip("XYZN", forward_vertices(d["sdf"], direction="front")))}),
    ("colorization", lambda d: {**d, "render": colorization(None, None, d["X"], d["Y"], d["Z"], d["calib"], d["N"],
                                                            resolution=bench.RESOLUTIONS[-1])}),
]
acc = {n: [] for n, _ in stages}
with torch.no_grad():
    for it in range(10):
        d = frame
        for name, fn in stages:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d = fn(d)
            torch.cuda.synchronize()
            if it >= 5:
                acc[name].append((time.perf_counter() - t0) * 1e3)
tot = 0.0
for name, _ in stages:
    m = float(np.median(acc[name]))
    tot += m
    print("%-18s %7.3f ms" % (name, m))
print("%-18s %7.3f ms" % ("sum", tot))
