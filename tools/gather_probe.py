"""Sampling stage alone (side build -DMP32_GATHER_ONLY): time and, under rocprofv3 --pmc, HBM-side bytes.
Points: the lattice nodes an octree level of the body fixture selects (spatially coherent) and
uniformly random points (no locality)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monoport_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmp_ablategather.so")
from monoport_amd import ops, synthetic as syn
dev = "cuda:0"
mlp = ops.PackedMLP.from_layers(dev, syn.body_mlp("G", noise=0.05, seed=1), 1)
fh = ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2))[None].to(dev))
cal = torch.eye(4, device=dev)[None]
n = 846080
# coherent: a thin shell of lattice nodes around a sphere, in z,y,x raster order like the octree's lists
r = 257
c = (np.arange(r, dtype=np.float32) + 0.5) / r * 2 - 1
zz, yy, xx = np.meshgrid(c, c, c, indexing="ij")
d = np.sqrt(xx ** 2 + yy ** 2 + zz ** 2)
sel = np.abs(d - 0.7) < 0.0163
shell = np.stack([xx[sel], yy[sel], zz[sel]]).astype(np.float32)[:, :n]
for name, p in (("coherent shell", shell), ("uniform random", syn.rand_points(shell.shape[1], 3, 1.0))):
    pt = torch.from_numpy(np.ascontiguousarray(p))[None].to(dev)
    for _ in range(3):
        ops.query(mlp, fh, pt, cal, syn.Z_SCALE)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.query(mlp, fh, pt, cal, syn.Z_SCALE)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%s: %d points, %.3f ms, %.2f TB/s gathered (4 taps x 1 KB per point)"
          % (name, pt.shape[2], ms, pt.shape[2] * 4096 / ms / 1e9))
