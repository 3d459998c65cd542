"""Run a few fused-query launches for a rocprofv3 --pmc pass (see profiles/)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoport_amd import synthetic as syn, ops
dev = "cuda:0"
mlp = ops.PackedMLP.from_layers(dev, syn.body_mlp("G", noise=0.05, seed=1), 1)
mlp.set_precision(os.environ.get("MP_PROBE_PREC", "f32"))
fh = ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2))[None].to(dev))
cal = torch.eye(4, device=dev)[None]
for n in [int(a) for a in sys.argv[1:]] or [64, 4913, 262144]:
    p = torch.from_numpy(syn.rand_points(n, 3, 1.0))[None].to(dev)
    for _ in range(3):
        ops.query(mlp, fh, p, cal, syn.Z_SCALE)
    torch.cuda.synchronize()
