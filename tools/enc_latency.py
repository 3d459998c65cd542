"""netG encoder time per frame by batch size: eager and as a hipGraph replay, hand-over dataflow
(round 3) vs round 2's per-module path.   python tools/enc_latency.py [precision] [batches...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from monoport_amd import synthetic as syn
from monoport_amd.modeling import backbones

dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
batches = [int(v) for v in sys.argv[2:]] or [1, 10]
backbones.set_encoder_conv_precision(prec)
net, _ = bench.build_netg(dev)
enc = net.image_filter


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    for b in batches:
        img = torch.stack([torch.from_numpy(syn.synthetic_image(i)) for i in range(b)]).to(dev)
        hwc = torch.empty((b, 128, 128, 256), device=dev)
        for flow in ("on", "on-1stream", "off"):
            backbones.ENCODER_DATAFLOW = "off" if flow == "off" else "on"
            backbones.ENCODER_BRANCHES = "off" if flow == "on-1stream" else "on"
            run = lambda: enc(img, last_only=True, hwc_out=hwc)
            t_eager = timed(run, 10)
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                run()
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    run()
                side.synchronize()
                t_graph = timed(graph.replay, 20)
            print("encoder %s batch %2d dataflow %-10s: eager %.3f ms/frame, graph %.3f ms/frame (%.3f ms per launch)"
                  % (prec, b, flow, t_eager / b, t_graph / b, t_graph), flush=True)
            del graph
