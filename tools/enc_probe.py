"""Time netG / netC encoders alone under a few PyTorch-ROCm settings."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from monoport_amd import synthetic as syn
dev = torch.device("cuda", 0)
net, _ = bench.build_netg(dev)
netc = bench.build_netc(dev)
img = torch.from_numpy(syn.synthetic_image(0))[None].to(dev)
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    print("netG encoder default: %.2f ms" % timeit(lambda: net.image_filter(img, last_only=True)))
    print("netC encoder default: %.2f ms" % timeit(lambda: netc.image_filter(img)))
    torch.backends.cudnn.benchmark = True
    print("netG encoder cudnn.benchmark: %.2f ms" % timeit(lambda: net.image_filter(img, last_only=True)))
    print("netC encoder cudnn.benchmark: %.2f ms" % timeit(lambda: netc.image_filter(img)))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        net.image_filter(img, last_only=True)
    s.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = net.image_filter(img, last_only=True)
    print("netG encoder graph replay: %.2f ms" % timeit(lambda: g.replay()))
