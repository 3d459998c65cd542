"""MIOpen fp32 3x3 convolutions of the hourglass shapes: NCHW vs channels_last."""
import time, torch, torch.nn.functional as F
dev = "cuda:0"
def bench(x, w, n=20):
    for _ in range(5): F.conv2d(x, w, padding=1)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): F.conv2d(x, w, padding=1)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for b in (4,):
    for (cin, cout, hw) in ((256, 128, 128), (128, 64, 128), (64, 64, 128), (256, 128, 64), (128, 64, 64), (256, 128, 32)):
        x = torch.randn(b, cin, hw, hw, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev)
        t0 = bench(x, w)
        xc = x.contiguous(memory_format=torch.channels_last); wc = w.contiguous(memory_format=torch.channels_last)
        t1 = bench(xc, wc)
        fl = 2 * b * cin * cout * 9 * hw * hw
        print("b%d %3d->%3d @%3d: NCHW %7.1f us (%5.1f TF)   NHWC %7.1f us (%5.1f TF)" % (b, cin, cout, hw, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6))
