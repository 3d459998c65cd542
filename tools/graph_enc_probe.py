"""Is a hipGraph of the ENCODER ONLY stable when tensors are allocated after capture?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from monoport_amd import synthetic as syn
from monoport_amd.recon import pifu_calib
dev = torch.device("cuda", 0)
net, _ = bench.build_netg(dev)
static_img = torch.zeros(1, 3, 512, 512, device=dev)
s = torch.cuda.Stream()
with torch.no_grad():
    with torch.cuda.stream(s):
        for _ in range(2):
            net.image_filter(static_img, last_only=True)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        feat = net.image_filter(static_img, last_only=True)[-1][0]
print("captured", flush=True)
images = [torch.from_numpy(syn.synthetic_image(i))[None].to(dev) for i in range(4)]
calibs = [pifu_calib(*syn.scene_camera(3 * i), device=dev) for i in range(23)]
junk = [torch.randn(1 << 20, device=dev) for _ in range(8)]
with torch.no_grad():
    for i in range(23):
        with torch.cuda.stream(s):
            static_img.copy_(images[i % 4])
            g.replay()
        s.synchronize()
        ref = net.image_filter(images[i % 4], last_only=True)[-1][0]
        torch.cuda.synchronize()
        assert torch.equal(ref, feat), i
print("23 replays after post-capture allocations: identical to eager", flush=True)
