import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from monoport_amd import synthetic as syn
dev = torch.device("cuda", 0)
net, _ = bench.build_netg(dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    for b in (1, 2, 3, 4, 8):
        img = torch.stack([torch.from_numpy(syn.synthetic_image(i)) for i in range(b)]).to(dev)
        ms = timeit(lambda: net.image_filter(img, last_only=True))
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            net.image_filter(img, last_only=True)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = net.image_filter(img, last_only=True)
        msg = timeit(lambda: g.replay())
        print("batch %d: eager %.2f ms (%.2f/frame)  graph %.2f ms (%.2f/frame)" % (b, ms, ms / b, msg, msg / b), flush=True)
