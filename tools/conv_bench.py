"""Per-shape time of the encoder's convolution kernels as the dataflow path launches them (statistics
hand-over + block tail in the epilogue), each measured as a hipGraph of back-to-back launches (no host
gaps).   python tools/conv_bench.py [batch ...]      MODES=auto,large1,large2,sk1,sk2 to choose variants"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from monoport_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda", 0)
batches = [int(v) for v in sys.argv[1:]] or [1, 10]
MODES = os.environ.get("MODES", "auto,large1,large2,sk1,sk2").split(",")
TUNE = {"auto": 0, "large1": 0x101, "large2": 0x102, "sk1": 0x201, "sk2": 0x202}
# (Cin, Cout, H=W, count per frame, Ctot of the block, offset)
SHAPES = [(256, 128, 128, 8, 256, 0), (128, 64, 128, 10, 256, 128), (64, 64, 128, 9, 256, 192),
          (128, 128, 128, 1, 256, 0), (64, 32, 128, 1, 128, 64), (32, 32, 128, 1, 128, 96),
          (256, 128, 64, 12, 256, 0), (128, 64, 64, 12, 256, 128), (64, 64, 64, 12, 256, 192),
          (256, 128, 32, 12, 256, 0), (128, 64, 32, 12, 256, 128), (64, 64, 32, 12, 256, 192),
          (64, 64, 256, 1, 128, 0), (64, 32, 256, 1, 128, 64), (32, 32, 256, 1, 128, 96)]
REPS = 20


def graph_time(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for _ in range(REPS):
                fn()
        side.synchronize()
        g.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        side.synchronize()
    return e0.elapsed_time(e1) / (3 * REPS) * 1e3  # us per launch


def main():
  with torch.no_grad():
      for b in batches:
          tot = {m: 0.0 for m in MODES}
          gf_tot = 0.0
          for cin, cout, hw, count, ctot, off in SHAPES:
              x = torch.randn((b, cin, hw, hw), device=dev)
              w = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
              packed = ops.PackedConv3x3(w)
              ss = torch.rand((b, cin, 2), device=dev)
              gn_x = torch.nn.GroupNorm(32, cin).to(dev)
              acc_x, acc_y, acc_o = (ops.gn_acc_zeros(dev, b) for _ in range(3))
              ident = torch.zeros((b, cin, 2), device=dev)
              ident[..., 0] = 1.0
              ops.gn_apply(x, ident, False, stats=acc_x)  # statistics of x, as its producer would leave them
              out = torch.empty((b, ctot, hw, hw), device=dev)
              res = torch.randn((b, ctot, hw, hw), device=dev)
              gf = 2.0 * 9 * cin * cout * hw * hw * b / 1e9
              line = "%3d->%3d @%3d^2 x%-2d (%5.1f GF):" % (cin, cout, hw, b, gf)
              best = None
              for m in MODES:
                  lib.mp_conv3x3_tune(TUNE[m])
                  try:
                      t = graph_time(lambda: ops.conv3x3_fused(x, (acc_x, gn_x), packed, stats=acc_y, out=out, res=res,
                                                               out_off=off, out_stats=acc_o))
                  except Exception as e:  # shape not served by this variant
                      t = float("nan")
                  lib.mp_conv3x3_tune(0)
                  line += "  %s %6.1f us %5.1f TF" % (m, t, gf / t * 1e3)
                  tot[m] += t * count / b
                  if m != "auto" and t == t and (best is None or t < best[1]):
                      best = (m, t)
              # the same launch without tail / statistics (round 2's kernel work)
              t_plain = graph_time(lambda: ops.conv3x3_gn(x, ss, packed, relu=True, want_stats=False))
              line += "  | plain %6.1f us" % t_plain
              if best:
                  line += "  best %s" % best[0]
              gf_tot += gf * count / b
              print(line, flush=True)
          print("batch %d, per frame (%.1f GFLOP): " % (b, gf_tot)
                + "  ".join("%s %.3f ms (%.1f TF)" % (m, tot[m] / 1e3, gf_tot / tot[m] * 1e3) for m in MODES), flush=True)
          # 1x1 convolutions of the hourglass tail
          for c2, res_on, stats, what in ((0, False, True, "conv_last"), (0, False, False, "l"), (256, True, True, "bl|al")):
              x1 = torch.randn((b, 256, 128, 128), device=dev)
              x2 = torch.randn((b, 256, 128, 128), device=dev) if c2 else None
              r = torch.randn((b, 256, 128, 128), device=dev) if res_on else None
              cv = torch.nn.Conv2d(256, 256, 1).to(dev)
              pk = ops.PackedConv1x1(cv.weight, cv.bias, cv.weight if c2 else None, cv.bias if c2 else None)
              gn = torch.nn.GroupNorm(32, 256).to(dev)
              acc_in, acc_out = (ops.gn_acc_zeros(dev, b) for _ in range(2))
              ident = torch.zeros((b, 256, 2), device=dev)
              ident[..., 0] = 1.0
              ops.gn_apply(x1, ident, False, stats=acc_in)
              gf = 2.0 * (256 + c2) * 256 * 128 * 128 * b / 1e9
              line = "1x1 %-9s x%-2d (%5.1f GF):" % (what, b, gf)
              for mrw in (0, 1, 2):
                  lib.mp_conv3x3_tune(mrw << 12)
                  t = graph_time(lambda: ops.conv1x1_fused(x1, (acc_in, gn), True, x2, pk, res=r,
                                                           stats=acc_out if stats else None))
                  lib.mp_conv3x3_tune(0)
                  line += "  mrw%d %6.1f us %5.1f TF" % (mrw, t, gf / t * 1e3)
              print(line, flush=True)


if __name__ == "__main__":
    main()
