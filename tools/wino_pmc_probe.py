"""A few launches of one 3x3 convolution shape through conv3x3_fused for a rocprofv3 --pmc pass.
    python tools/wino_pmc_probe.py <tune: 0 auto | 0x800 k64 | 0x1000 k128 | 0x400 direct> [cin cout hw batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoport_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda", 0)
tune = int(sys.argv[1], 0)
cin, cout, hw, b = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (256, 128, 128, 20)
with torch.no_grad():
    x = torch.randn((b, cin, hw, hw), device=dev)
    w = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
    packed = ops.PackedConv3x3(w)
    gn_x = torch.nn.GroupNorm(32, cin).to(dev)
    ident = torch.zeros((b, cin, 2), device=dev); ident[..., 0] = 1.0
    acc_x = ops.gn_acc_zeros(dev, b)
    ops.gn_apply(x, ident, False, stats=acc_x)
    res = torch.randn((b, 256, hw, hw), device=dev)
    out = torch.zeros((b, 256, hw, hw), device=dev)
    lib.mp_conv3x3_tune(tune)
    for _ in range(4):
        a2, a3 = ops.gn_acc_zeros(dev, b), ops.gn_acc_zeros(dev, b)
        ops.conv3x3_fused(x, (acc_x, gn_x), packed, stats=a2, out=out, res=res, out_off=0, out_stats=a3)
    torch.cuda.synchronize()
