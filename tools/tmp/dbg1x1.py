import sys, torch
sys.path.insert(0, "/root/repo")
from monoport_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
n, c, hw = 2, 256, 64
y = torch.randn((n, c, hw, hw), generator=g).to(dev)
w = (torch.randn((256, c, 1, 1), generator=g) * 0.05).to(dev)
b = torch.randn((256,), generator=g).to(dev)
p = ops.PackedConv1x1(w, b)
t, st = ops.conv1x1(y, None, False, None, p, want_stats=True)
ref = torch.nn.functional.conv2d(y.double(), w.double(), b.double())
err = (t.double() - ref).abs()
print("max err", err.max().item())
pc = err.amax(dim=(0, 2, 3))
bad = (pc > 1e-3).nonzero().flatten().tolist()
print("bad channels", bad[:64], len(bad))
pp = err.amax(dim=(0, 1)).flatten()
badp = (pp > 1e-3).nonzero().flatten().tolist()
print("bad pixels", badp[:40], len(badp))
if bad:
    ch = bad[0]
    print("t - ref on bad channel, first pixels", (t.double() - ref)[0, ch].flatten()[:8].tolist(), "bias", b[ch].item())
