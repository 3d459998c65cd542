"""Winograd F(2x2, 3x3) kernel (csrc/conv_wino.hip) against the direct kernel and the fp64 convolution, and its time
per shape next to the direct kernel's (hipGraph of back-to-back launches).   python tools/wino_check.py [batch ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from monoport_amd import _lib, ops
from tools.conv_bench import graph_time

lib = _lib.load()
dev = torch.device("cuda", 0)
batches = [int(v) for v in sys.argv[1:]] or [1, 20]
SHAPES = [(256, 128, 128, 256, 0), (256, 128, 64, 256, 0), (256, 128, 32, 256, 0), (128, 128, 128, 256, 0),
          (128, 64, 128, 256, 128), (64, 64, 128, 256, 192), (128, 64, 64, 256, 128), (64, 64, 64, 256, 192),
          (128, 64, 32, 256, 128), (64, 64, 32, 256, 192)]
with torch.no_grad():
    for b in batches:
        for cin, cout, hw, ctot, off in SHAPES:
            g = torch.Generator().manual_seed(cin + hw)
            x = (torch.randn((b, cin, hw, hw), generator=g) * 2 + 0.3).to(dev)
            w = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
            packed = ops.PackedConv3x3(w)
            gn_x = torch.nn.GroupNorm(32, cin).to(dev)
            gn_x.weight.copy_(torch.rand(cin, generator=g) + 0.5)
            gn_x.bias.copy_(torch.rand(cin, generator=g) - 0.5)
            ident = torch.zeros((b, cin, 2), device=dev)
            ident[..., 0] = 1.0
            acc_x = ops.gn_acc_zeros(dev, b)
            ops.gn_apply(x, ident, False, stats=acc_x)
            res = torch.randn((b, ctot, hw, hw), generator=g).to(dev)
            outs = {}
            times = {}
            for mode, tune in (("wino", 0), ("wino64", 0x800), ("direct", 0x400)):
                lib.mp_conv3x3_tune(tune)
                out = torch.zeros((b, ctot, hw, hw), device=dev)
                acc_y, acc_o = ops.gn_acc_zeros(dev, b), ops.gn_acc_zeros(dev, b)
                y = ops.conv3x3_fused(x, (acc_x, gn_x), packed, stats=acc_y, out=out, res=res, out_off=off, out_stats=acc_o)
                torch.cuda.synchronize()
                outs[mode] = (y, out, acc_y.clone(), acc_o.clone())
                a2, a3 = ops.gn_acc_zeros(dev, b), ops.gn_acc_zeros(dev, b)
                times[mode] = graph_time(lambda: ops.conv3x3_fused(x, (acc_x, gn_x), packed, stats=a2, out=out, res=res,
                                                                   out_off=off, out_stats=a3))
                lib.mp_conv3x3_tune(0)
            nb = min(b, 2)
            ss = ops.gn_reference_ss(acc_x, gn_x, (cin // 32) * hw * hw).double()
            v = torch.relu(x[:nb].double() * ss[:nb, :, 0, None, None] + ss[:nb, :, 1, None, None])
            ref = torch.nn.functional.conv2d(v, w.double(), padding=1)
            scale = max(1.0, ref.abs().max().item())
            ew = (outs["wino"][0][:nb].double() - ref).abs().max().item()
            ed = (outs["direct"][0][:nb].double() - ref).abs().max().item()
            eo = (outs["wino"][1] - outs["direct"][1]).abs().max().item()
            sy = (ops.gn_reference_ss(outs["wino"][2], torch.nn.GroupNorm(32, cout).to(dev), (cout // 32) * hw * hw)
                  - ops.gn_reference_ss(outs["direct"][2], torch.nn.GroupNorm(32, cout).to(dev), (cout // 32) * hw * hw)).abs().max().item()
            gf = 2.0 * 9 * cin * cout * hw * hw * b / 1e9
            e64 = (outs["wino64"][0][:nb].double() - ref).abs().max().item()
            eo64 = (outs["wino64"][1] - outs["direct"][1]).abs().max().item()
            print("%3d->%3d @%3d^2 x%-2d: wino %7.1f us (%5.1f TF-eq)  wino64 %7.1f us (%5.1f)  direct %7.1f us (%5.1f) | max|d| vs fp64: %.2e / %.2e / %.2e "
                  "(scale %.1f); tail vs direct %.2e / %.2e; next-GN ss %.2e"
                  % (cin, cout, hw, b, times["wino"], gf / times["wino"] * 1e3, times["wino64"], gf / times["wino64"] * 1e3,
                     times["direct"], gf / times["direct"] * 1e3, ew, e64, ed, scale, eo, eo64, sy), flush=True)
