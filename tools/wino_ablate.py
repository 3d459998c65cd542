"""Timing of the Winograd kernels under side builds of csrc/conv_wino.hip that leave work out (wrong results):
   python tools/wino_ablate.py <side library>    -- both kernels on 256 -> 128 at 128^2 x 20 and 128 -> 64 at 128^2 x 20"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoport_amd import _lib
LIBNAME = sys.argv[1] if len(sys.argv) > 1 else "product"
sys.argv = sys.argv[:1]
if LIBNAME != "product":
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "side", LIBNAME)
from monoport_amd import ops
from tools.conv_bench import graph_time
lib = _lib.load()
dev = torch.device("cuda", 0)
b = 20
line = "%-28s" % LIBNAME
with torch.no_grad():
    for cin, cout, hw, tune in ((256, 128, 128, 0x1000), (256, 128, 128, 0x800), (128, 64, 128, 0), (64, 64, 128, 0)):
        x = torch.randn((b, cin, hw, hw), device=dev)
        w = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
        packed = ops.PackedConv3x3(w)
        gn_x = torch.nn.GroupNorm(32, cin).to(dev)
        ident = torch.zeros((b, cin, 2), device=dev); ident[..., 0] = 1.0
        acc_x = ops.gn_acc_zeros(dev, b)
        ops.gn_apply(x, ident, False, stats=acc_x)
        res = torch.randn((b, 256, hw, hw), device=dev)
        out = torch.zeros((b, 256, hw, hw), device=dev)
        a2, a3 = ops.gn_acc_zeros(dev, b), ops.gn_acc_zeros(dev, b)
        lib.mp_conv3x3_tune(tune)
        t = graph_time(lambda: ops.conv3x3_fused(x, (acc_x, gn_x), packed, stats=a2, out=out, res=res, out_off=0, out_stats=a3))
        lib.mp_conv3x3_tune(0)
        line += "  %d->%d %s %6.1f us" % (cin, cout, {0x1000: "k128", 0x800: "k64", 0: "k64"}[tune], t)
print(line, flush=True)
