import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoport_amd import _lib, ops
dev = torch.device("cuda", 0)
n, cin, cout, h, w, ctot, off = 1, 256, 128, 128, 128, 256, 0
g = torch.Generator().manual_seed(1)
x = (torch.randn((n, cin, h, w), generator=g) * 2 + 0.3).to(dev)
res = torch.randn((n, ctot, h, w), generator=g).to(dev)
wt = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
packed = ops.PackedConv3x3(wt)
out = torch.full((n, ctot, h, w), 7.0, device=dev)
acc_y, acc_o = ops.gn_acc_zeros(dev, n), ops.gn_acc_zeros(dev, n)
y = ops.conv3x3_fused(x, None, packed, relu=False, stats=acc_y, out=out, res=res, out_off=off, out_stats=acc_o)
torch.cuda.synchronize()
want = y + res[:, off:off + cout]
bad = (out[:, off:off + cout] != want)
print("mismatches", int(bad.sum()), "of", bad.numel(), "max diff", (out[:, off:off+cout] - want).abs().max().item())
idx = bad.nonzero()
print(idx[:20].tolist())
print("by x parity", [int(bad[..., p::2].sum()) for p in (0, 1)], "by y parity", [int(bad[:, :, p::2].sum()) for p in (0, 1)])
print("by channel%32 (first 32)", [int(bad[:, c::32].sum()) for c in range(32)])
print("untouched ok", bool((out[:, cout:] == 7.0).all()))
o = out[0, 5, 10, :8].tolist(); yy = y[0, 5, 10, :8].tolist(); rr = res[0, 5, 10, :8].tolist()
print("out", [round(v, 4) for v in o]); print("y  ", [round(v, 4) for v in yy]); print("res", [round(v, 4) for v in rr])
print("y+res", [round(a + b, 4) for a, b in zip(yy, rr)])
