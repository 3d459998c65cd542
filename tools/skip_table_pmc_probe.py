"""A few launches of the skip-table query path for a rocprofv3 --pmc pass: the 16 tables of a batch
(skip_table_kernel) and one 885 k-point lattice launch of pifu_query_tab_kernel, three times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monoport_amd import _lib  # noqa: E402
from monoport_amd import ops, synthetic as syn  # noqa: E402
from monoport_amd.recon import pifu_calib  # noqa: E402
from skip_table_probe import lattice_points  # noqa: E402

dev = torch.device("cuda", 0)
mlp = ops.PackedMLP.from_layers(dev, syn.body_mlp("G", noise=0.05, seed=1), 1)
feats = torch.stack([ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2 + i))[None].to(dev))
                     for i in range(16)])
cal = pifu_calib(*syn.scene_camera(30), device=dev)
p = torch.from_numpy(lattice_points())[None].to(dev)
for _ in range(3):
    handle = ops.skip_table_batch(mlp, feats)
    ops.query(mlp, feats[0], p, cal, syn.Z_SCALE)
    torch.cuda.synchronize()
