import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monoport_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "full":
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmp_ablate%s.so" % sys.argv[1])
from monoport_amd import synthetic as syn, ops
dev = "cuda:0"
mlp = ops.PackedMLP.from_layers(dev, syn.body_mlp("G", noise=0.05, seed=1), 1); mlp.set_precision("f16x3")
fh = ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2))[None].to(dev)); cal = torch.eye(4, device=dev)[None]
for n in (128, 262144):
    pt = torch.from_numpy(syn.rand_points(n, 3, 1.0))[None].to(dev)
    for _ in range(2): ops.query(mlp, fh, pt, cal, syn.Z_SCALE)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.query(mlp, fh, pt, cal, syn.Z_SCALE)
    e1.record(); torch.cuda.synchronize()
    print(sys.argv[1:], "N=%d: %.3f ms" % (n, e0.elapsed_time(e1) / 5))
