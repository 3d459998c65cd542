"""Timing experiments on side builds of the library (never the product).

  python tools/ablate.py build NAME FILE.hip[:STEM] -DFLAG [-DFLAG ...]   # here: lib/libmp_ablateNAME.so
  python tools/ablate.py run NAME [precision] [n ...]              # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(name, src, flags):
    """``src`` may be FILE.hip:STEM -- a side source (e.g. an older revision saved next to the
    product's) that REPLACES the product's STEM.hip in the side library."""
    from monoport_amd import build as b
    lib_dir = os.path.join(ROOT, "monoport_amd", "lib")
    obj = os.path.join(lib_dir, "obj", "ablate_%s.o" % name)
    stem = os.path.splitext(os.path.basename(src))[0]
    if ":" in src:
        src, stem = src.split(":")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.FLAGS + flags +
                          ["-c", os.path.join(ROOT, "monoport_amd", "csrc", src), "-o", obj])
    others = [os.path.join(lib_dir, "obj", s.replace(".hip", ".o")) for s in b.SOURCES
              if not s.startswith(stem + ".")]
    out = os.path.join(lib_dir, "libmp_ablate%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out,
                           obj] + others)
    print(out)


def run(name, precision, sizes):
    import torch
    from monoport_amd import _lib
    if name != "full":
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmp_ablate%s.so" % name)
    from monoport_amd import ops, synthetic as syn
    dev = "cuda:0"
    mlp = ops.PackedMLP.from_layers(dev, syn.body_mlp("G", noise=0.05, seed=1), 1)
    mlp.set_precision(precision)
    fh = ops.pack_features(torch.from_numpy(syn.body_feat(256, 128, 128, 2))[None].to(dev))
    cal = torch.eye(4, device=dev)[None]
    for n in sizes:
        pt = torch.from_numpy(syn.rand_points(n, 3, 1.0))[None].to(dev)
        for _ in range(3):
            ops.query(mlp, fh, pt, cal, syn.Z_SCALE)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if n < 100000 else 5
        e0.record()
        for _ in range(reps):
            ops.query(mlp, fh, pt, cal, syn.Z_SCALE)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-10s %-6s N=%-8d %.3f ms  %.1f TFLOP/s-equivalent" % (name, precision, n, ms,
              n * 2363906 / ms / 1e9))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
        sizes = [int(v) for v in sys.argv[4:]] or [64, 4913, 9086, 25098, 50860, 211484, 1048576]
        run(sys.argv[2], prec, sizes)
