"""Accuracy + speed of the f16x3 query kernel against the fp64 oracle and the f32 kernel."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from monoport_amd import synthetic as syn, ops
from oracle import pifu_oracle as orc
dev = "cuda:0"
for name, layers, f in (("rand", syn.rand_mlp("G", 11, 2.0), syn.rand_feat(256, 128, 128, 21)),
                        ("body", syn.body_mlp("G", noise=0.05, seed=13), syn.body_feat(256, 128, 128, 23))):
    p = syn.rand_points(5000, 31, 1.2)
    calib = orc.pifu_calib(*syn.scene_camera(33))
    mlp = ops.PackedMLP.from_layers(dev, layers, 1)
    fh = ops.pack_features(torch.from_numpy(f)[None].to(dev))
    pt = torch.from_numpy(p)[None].to(dev); cal = torch.from_numpy(calib).to(dev)
    o32 = ops.query(mlp, fh, pt, cal, syn.Z_SCALE)[0].cpu().numpy()
    mlp.set_precision("f16x3")
    o16 = ops.query(mlp, fh, pt, cal, syn.Z_SCALE)[0].cpu().numpy()
    ref = orc.query(f, p, calib[0], layers, 1, syn.Z_SCALE, precision="f64")
    print(name, "|f32-f64| %.3g  |f16x3-f64| %.3g  |f16x3-f32| %.3g  nan=%d" % (
        np.abs(o32 - ref).max(), np.abs(o16 - ref).max(), np.abs(o16 - o32).max(), np.isnan(o16).sum()))
for n in (4913, 65536, 262144, 1048576):
    pt = torch.from_numpy(syn.rand_points(n, 3, 1.0))[None].to(dev)
    for _ in range(2): ops.query(mlp, fh, pt, cal, syn.Z_SCALE)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.query(mlp, fh, pt, cal, syn.Z_SCALE)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
    print("f16x3 query N=%d: %.3f ms  %.1f Mpts/s  %.0f TFLOP/s-equivalent" % (n, ms, n / ms / 1e3, n * 2363906 / ms / 1e9))
