"""netG.filter at batch 1 / 2 as the drop-in surface calls it: launch by launch, as a recorded mp_plan (csrc/plan.hip)
and as a hipGraph -- back-to-back throughput (host never waits) and single-call latency (synchronised around each call),
full API (4 stage outputs) and last_only + channels-last output.   python tools/enc_plan_probe.py [batches...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from monoport_amd import synthetic as syn
from monoport_amd.modeling import backbones

dev = torch.device("cuda", 0)
batches = [int(v) for v in sys.argv[1:]] or [1, 2]
net, _ = bench.build_netg(dev)
enc = net.image_filter


def stream_ms(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / reps * 1e3
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, host


def latency_ms(fn, reps=15):
    for _ in range(3):
        fn()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    out.sort()
    return out[len(out) // 2]


with torch.no_grad():
    for b in batches:
        img = torch.stack([torch.from_numpy(syn.synthetic_image(i)) for i in range(b)]).to(dev)
        hwc = torch.empty((b, 128, 128, 256), device=dev)
        for what, call in (("filter (4 stage outputs)", lambda g: enc(img, graphed=g)),
                           ("last_only + hwc_out", lambda g: enc(img, last_only=True, hwc_out=hwc, graphed=g))):
            for mode in ("eager", "plan", "plan-1stream", "graph"):
                backbones.ENCODER_PLAN = "on" if mode.startswith("plan") else "off"
                backbones.ENCODER_BRANCHES = "off" if mode == "plan-1stream" else "on"
                g = False if mode == "eager" else True if mode == "graph" else None
                fn = lambda: call(g)
                gpu, host = stream_ms(fn)
                lat = latency_ms(fn)
                print("batch %d %-26s %-12s: back to back %.3f ms per call (host %.3f ms), single call %.3f ms"
                      % (b, what, mode, gpu, host, lat), flush=True)
