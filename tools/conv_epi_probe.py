"""What the pieces of the fused 3x3 epilogue cost: plain / partial sums only / hand-over / block tail / all.
   python tools/conv_epi_probe.py [batch ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from monoport_amd import ops
from conv_bench import graph_time, dev  # noqa: E402

batches = [int(v) for v in sys.argv[1:]] or [1, 10]
SHAPES = [(256, 128, 128, 256, 0), (128, 64, 128, 256, 128), (64, 64, 128, 256, 192), (256, 128, 64, 256, 0),
          (64, 64, 64, 256, 192), (64, 64, 32, 256, 192)]
with torch.no_grad():
    for b in batches:
        for cin, cout, hw, ctot, off in SHAPES:
            x = torch.randn((b, cin, hw, hw), device=dev)
            w = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
            packed = ops.PackedConv3x3(w)
            ss = torch.rand((b, cin, 2), device=dev)
            gn_x = torch.nn.GroupNorm(32, cin).to(dev)
            acc_x, acc_y, acc_o = (ops.gn_acc_zeros(dev, b) for _ in range(3))
            ident = torch.zeros((b, cin, 2), device=dev)
            ident[..., 0] = 1.0
            ops.gn_apply(x, ident, False, stats=acc_x)
            gin = (acc_x, gn_x)
            out = torch.empty((b, ctot, hw, hw), device=dev)
            res = torch.randn((b, ctot, hw, hw), device=dev)
            t = {}
            t["plain(ss)"] = graph_time(lambda: ops.conv3x3_gn(x, ss, packed, relu=True, want_stats=False))
            t["plain(acc)"] = graph_time(lambda: ops.conv3x3_fused(x, gin, packed))
            t["partial"] = graph_time(lambda: ops.conv3x3_gn(x, ss, packed, relu=True, want_stats=True))
            t["stats"] = graph_time(lambda: ops.conv3x3_fused(x, gin, packed, stats=acc_y))
            t["tail"] = graph_time(lambda: ops.conv3x3_fused(x, gin, packed, out=out, res=res, out_off=off))
            t["tail+st2"] = graph_time(lambda: ops.conv3x3_fused(x, gin, packed, out=out, res=res, out_off=off,
                                                                 out_stats=acc_o))
            t["all"] = graph_time(lambda: ops.conv3x3_fused(x, gin, packed, stats=acc_y, out=out, res=res,
                                                            out_off=off, out_stats=acc_o))
            print("%3d->%3d @%3d^2 x%-2d: " % (cin, cout, hw, b) + "  ".join("%s %.1f" % kv for kv in t.items()), flush=True)
