"""Per-shape timing of csrc/conv3x3.hip against MIOpen's fp32 convolution (+ the stand-alone
GroupNorm kernel it needs in front) at the encoder's shapes.   python tools/conv_probe.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from monoport_amd import _lib, ops

lib = _lib.load()

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
# (Cin, Cout, H=W, count per frame) of every 3x3 convolution of HGFilter (4 stacks, depth 2)
SHAPES = [(256, 128, 128, 8), (128, 64, 128, 10), (64, 64, 128, 9), (128, 128, 128, 1), (64, 32, 128, 1),
          (32, 32, 128, 1), (256, 128, 64, 12), (128, 64, 64, 12), (64, 64, 64, 12), (256, 128, 32, 12),
          (128, 64, 32, 12), (64, 64, 32, 12), (64, 64, 256, 1), (64, 32, 256, 1), (32, 32, 256, 1)]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot_h = tot_m = tot_f = 0.0
with torch.no_grad():
    for cin, cout, hw, count in SHAPES:
        x = torch.randn((B, cin, hw, hw), device=dev)
        w = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
        gn = torch.nn.GroupNorm(32, cin).to(dev)
        packed = ops.PackedConv3x3(w)
        ss = ops.gn_finalize(ops.gn_stats(x, 32), B, cin, 32, (cin // 32) * hw * hw, gn.weight, gn.bias, gn.eps)
        t_nr = {}
        for nr in (4, 2, 1):
            lib.mp_conv3x3_tune(nr)
            t_nr[nr] = timed(lambda: ops.conv3x3_gn(x, ss, packed, relu=True, want_stats=True))
        lib.mp_conv3x3_tune(0)
        t_hip = timed(lambda: ops.conv3x3_gn(x, ss, packed, relu=True, want_stats=True))
        packed16 = ops.PackedConv3x3(w, "f16x3")
        t16 = {}
        for nr in (4, 2, 1, 0):
            lib.mp_conv3x3_tune(nr)
            t16[nr] = timed(lambda: ops.conv3x3_gn(x, ss, packed16, relu=True, want_stats=True))
        tot_16 = globals().get("tot_16", 0.0) + t16[0] * count / B
        globals()["tot_16"] = tot_16
        v = ops.group_norm(x, 32, gn.weight, gn.bias, gn.eps, relu=True)
        t_mi = timed(lambda: F.conv2d(v, w, padding=1))
        t_gn = timed(lambda: ops.group_norm(x, 32, gn.weight, gn.bias, gn.eps, relu=True))
        gf = 2.0 * 9 * cin * cout * hw * hw * B / 1e9
        print("   NR 4/2/1: %.3f / %.3f / %.3f ms | f16x3 NR 4/2/1/auto: %.3f / %.3f / %.3f / %.3f ms = %.0f TFLOP/s-eq"
              % (t_nr[4], t_nr[2], t_nr[1], t16[4], t16[2], t16[1], t16[0], 2.0 * 9 * cin * cout * hw * hw * B / 1e9 / t16[0]))
        print("%4d -> %3d @ %3d^2 x%d: hip %.3f ms = %6.1f TFLOP/s | MIOpen %.3f ms = %6.1f TFLOP/s (+ GroupNorm "
              "pass %.3f ms) | x%d per frame" % (cin, cout, hw, B, t_hip, gf / t_hip, t_mi, gf / t_mi, t_gn, count))
        tot_h += t_hip * count / B
        tot_m += (t_mi + t_gn) * count / B
        tot_f += gf * count / B
print("per frame, all 3x3 convolutions (%.1f GFLOP): hip f32 %.3f ms (%.1f TFLOP/s) | hip f16x3 %.3f ms | MIOpen + GroupNorm "
      "passes %.3f ms" % (tot_f, tot_h, tot_f / tot_h, globals().get("tot_16", 0.0), tot_m))
