"""Encoder-only workload (batch 4, 12 eager passes) for a rocprofv3 --kernel-trace --stats run."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from monoport_amd import synthetic as syn
dev = torch.device("cuda", 0)
net, _ = bench.build_netg(dev)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
img = torch.stack([torch.from_numpy(syn.synthetic_image(i)) for i in range(b)]).to(dev)
with torch.no_grad():
    for _ in range(12):
        net.image_filter(img, last_only=True)
torch.cuda.synchronize()
