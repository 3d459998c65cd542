"""Does giving the encoder stage and the reconstruction stage disjoint shares of the CUs let them run side by
side?  (round 6; VERDICT r5 item 2 -- the reference's stage threads put netG.filter of frame k+1 and reconEngine of
frame k in flight together, RTL/main.py:366-395, RTL/dataloader.py:1026-1053)

  python tools/cu_partition_probe.py        (on the GPU box)

1. what a CU mask means on this device: mp_mfma_clock_probe on streams of mp_stream_create_cu_mask -- TFLOP/s
   against the share of the mask, alone and two complementary masks at once;
2. netG.filter at batch 1 (launch by launch, one stream) by CUs; one frame's skip table + 17..257 reconstruction by CUs;
3. both in a loop from two host threads: ordinary streams against complementary CU masks, frames per second.
"""
import ctypes
import os
import sys
import threading
import time

os.environ.setdefault("MONOPORT_ENCODER_PLAN", "off")
os.environ.setdefault("MONOPORT_ENCODER_BRANCHES", "off")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from monoport_amd import ops, synthetic as syn
from monoport_amd.recon import pifu_calib

dev = torch.device("cuda", 0)
ctx = ops.get_context(dev)
ctx2 = ops.Context(0)  # a second context: mp_mfma_clock_probe holds its context's mutex while it runs
lib = ctx.lib
N_CU = lib.mp_stream_cu_count(ctx.handle, None)


def masked(c, first, n):
    h = ctypes.c_void_p()
    c.check(lib.mp_stream_create_cu_mask(c.handle, first, n, ctypes.byref(h)), "mp_stream_create_cu_mask")
    return h


def probe(c, st, ms=30.0):
    out = (ctypes.c_double * 4)()
    c.check(lib.mp_mfma_clock_probe(c.handle, ctypes.c_float(ms), out, st), "mp_mfma_clock_probe")
    return out[0], out[1], out[2]


print("device CUs:", N_CU)
full = probe(ctx, None)
print("whole device: %.1f TFLOP/s at %.0f MHz" % full[:2])
for first, n in ((0, 32), (0, 64), (0, 128), (0, 192), (64, 192), (128, 128), (32, 32), (0, 8), (0, 16)):
    st = masked(ctx, first, n)
    t = probe(ctx, st)
    print("mask [%3d, %3d): %6.1f TFLOP/s = %.3f of the device for %.3f of its CUs (%.0f MHz, %.1f ms)"
          % (first, first + n, t[0], t[0] / full[0], n / N_CU, t[1], t[2]))
    ctx.check(lib.mp_stream_destroy(ctx.handle, st), "mp_stream_destroy")
for na in (64, 96):
    sa, sb = masked(ctx, 0, na), masked(ctx2, na, N_CU - na)
    res = {}
    th = [threading.Thread(target=lambda k=k, c=c, s=s: res.__setitem__(k, probe(c, s, 200.0)))
          for k, c, s in (("a", ctx, sa), ("b", ctx2, sb))]
    [t.start() for t in th]
    [t.join() for t in th]
    print("complementary masks at once: [0,%d) %.1f TFLOP/s + [%d,%d) %.1f TFLOP/s = %.1f (whole device alone %.1f)"
          % (na, res["a"][0], na, N_CU, res["b"][0], res["a"][0] + res["b"][0], full[0]))
    sc = masked(ctx2, 0, na)  # the SAME CUs from two streams: they must share
    th = [threading.Thread(target=lambda k=k, c=c, s=s: res.__setitem__(k, probe(c, s, 200.0)))
          for k, c, s in (("a", ctx, sa), ("b", ctx2, sc))]
    [t.start() for t in th]
    [t.join() for t in th]
    print("the same mask [0,%d) twice at once: %.1f + %.1f TFLOP/s" % (na, res["a"][0], res["b"][0]))
    for c, s in ((ctx, sa), (ctx2, sb), (ctx2, sc)):
        c.check(lib.mp_stream_destroy(c.handle, s), "mp_stream_destroy")

# ---- the two stages --------------------------------------------------------------------------------------
netG, _ = bench.build_netg(dev)
img = torch.from_numpy(syn.synthetic_image(0))[None].to(dev)
planes = torch.from_numpy(syn.body_feature_planes(128, 128)).to(dev)
calib = pifu_calib(*syn.scene_camera(30), device=dev)
mlp = netG.surface_classifier.packed()
RES = bench.RESOLUTIONS
with torch.no_grad():
    feats = netG.filter(img)
feats[-1][0][0, 0:2].copy_(planes)
fh = ops.pack_features(feats[-1][0])
torch.cuda.synchronize()


def encoder_once():
    with torch.no_grad():
        return netG.filter(img)


tables = {}


def recon_once():
    buf = tables.get("buf")
    if buf is None:  # one table buffer, rewritten per frame like a pipeline slot's
        buf = tables["buf"] = torch.empty((128, 128, ops.SKIP_TABLE_ROWS), dtype=torch.float32, device=dev)
    tables["handle"] = ops.skip_table(mlp, fh, out=buf)
    return ops.recon(mlp, fh, calib, syn.Z_SCALE, bench.B_MIN, bench.B_MAX, RES)


def timed(fn, stream, n=30, warm=5):
    with torch.cuda.stream(stream):
        for _ in range(warm):
            fn()
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        stream.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def ext(h):
    return torch.cuda.ExternalStream(h.value, device=dev)


plain = torch.cuda.Stream(device=dev)
print("netG.filter batch 1, one stream, launch by launch: %.3f ms on an ordinary stream" % timed(encoder_once, plain))
for n in (128, 96, 64, 48, 32):
    h = masked(ctx, 0, n)
    print("   %3d CUs: %.3f ms" % (n, timed(encoder_once, ext(h))))
    torch.cuda.synchronize()
    ctx.check(lib.mp_stream_destroy(ctx.handle, h), "mp_stream_destroy")
print("skip table + 17..257 reconstruction of one frame: %.3f ms on an ordinary stream" % timed(recon_once, plain))
for n in (224, 208, 192, 160, 128):
    h = masked(ctx, N_CU - n, n)
    print("   %3d CUs: %.3f ms" % (n, timed(recon_once, ext(h))))
    torch.cuda.synchronize()
    ctx.check(lib.mp_stream_destroy(ctx.handle, h), "mp_stream_destroy")


def both(se, sr, n=60):
    done = {}

    def loop(name, fn, st):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(st):
            for _ in range(5):
                fn()
            st.synchronize()
            bar.wait()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            st.synchronize()
            done[name] = time.perf_counter() - t0

    bar = threading.Barrier(2)
    th = [threading.Thread(target=loop, args=a) for a in (("enc", encoder_once, se), ("rec", recon_once, sr))]
    [t.start() for t in th]
    [t.join() for t in th]
    return done["enc"] / n * 1e3, done["rec"] / n * 1e3


e, r = both(torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
print("both stages looping from two host threads, ordinary streams: encoder %.3f ms, recon %.3f ms per frame -> %.1f frames/s"
      % (e, r, 1e3 / max(e, r)))
for ne in (32, 48, 64, 80, 96, 128):
    he, hr = masked(ctx, 0, ne), masked(ctx, ne, N_CU - ne)
    e, r = both(ext(he), ext(hr))
    print("   encoder on %3d CUs | recon on %3d: encoder %.3f ms, recon %.3f ms per frame -> %.1f frames/s"
          % (ne, N_CU - ne, e, r, 1e3 / max(e, r)))
    torch.cuda.synchronize()
    for h in (he, hr):
        ctx.check(lib.mp_stream_destroy(ctx.handle, h), "mp_stream_destroy")
