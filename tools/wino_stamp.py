"""Side build of csrc/conv_wino.hip with -DWN_STAMP (monoport_amd/lib/side/libmonoport_stamp.so): cycles a workgroup
spends in its prologue / K loop / output transform / epilogue (s_memtime, 100 MHz ticks x 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monoport_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "side", "libmonoport_stamp.so")
from monoport_amd import ops
dev = torch.device("cuda", 0)
with torch.no_grad():
    for b, cin, cout, hw in ((20, 256, 128, 128), (20, 128, 128, 128)):
        x = torch.randn((b, cin, hw, hw), device=dev)
        w = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
        packed = ops.PackedConv3x3(w)
        gn_x = torch.nn.GroupNorm(32, cin).to(dev)
        ident = torch.zeros((b, cin, 2), device=dev); ident[..., 0] = 1.0
        acc_x = ops.gn_acc_zeros(dev, b)
        ops.gn_apply(x, ident, False, stats=acc_x)
        res = torch.randn((b, 256, hw, hw), device=dev)
        out = torch.zeros((b, 256, hw, hw), device=dev)
        for _ in range(3):
            a2, a3 = ops.gn_acc_zeros(dev, b), ops.gn_acc_zeros(dev, b)
            y = ops.conv3x3_fused(x, (acc_x, gn_x), packed, stats=a2, out=out, res=res, out_off=0, out_stats=a3)
        torch.cuda.synchronize()
        # one stamp set per workgroup: at (img, channel 0, y0, x0 .. x0 + 3) of every 8 x 16 tile block
        st = y[:, 0].reshape(b, hw // 8, 8, hw // 16, 16)[:, :, 0, :, 0:4].reshape(-1, 4).double()
        print("x%d %d->%d @%d^2: %d workgroups; mean ticks (100 MHz): prologue %.0f  K loop %.0f  output transform %.0f  epilogue %.0f | us: %.1f %.1f %.1f %.1f"
              % (b, cin, cout, hw, st.shape[0], *st.mean(0).tolist(), *(st.mean(0) / 100.0).tolist()))
