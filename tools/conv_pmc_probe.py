"""A few launches of the 256->128 @128^2 x5 convolution for a rocprofv3 --pmc pass.
    python tools/conv_pmc_probe.py [f32|f16x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoport_amd import ops
dev = "cuda:0"
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
x = torch.randn((5, 256, 128, 128), device=dev)
w = torch.randn((128, 256, 3, 3), device=dev) * 0.05
gn = torch.nn.GroupNorm(32, 256).to(dev)
packed = ops.PackedConv3x3(w, prec)
ss = ops.gn_finalize(ops.gn_stats(x, 32), 5, 256, 32, 8 * 128 * 128, gn.weight, gn.bias, gn.eps)
for _ in range(4):
    ops.conv3x3_gn(x, ss, packed, relu=True, want_stats=True)
torch.cuda.synchronize()
