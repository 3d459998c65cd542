"""ms per frame of netC's image encoder (ResnetFilter) on the stock ops vs csrc/conv3x3.hip.

  python tools/netc_encoder_probe.py            (on the GPU box)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from monoport_amd import synthetic as syn
from monoport_amd.modeling import backbones

dev = torch.device("cuda", 0)
net = backbones.ResnetFilter().eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(shapes, 5).items()})
net.to(dev)
img = torch.from_numpy(syn.synthetic_image(6))[None].to(dev)
for conv, prec in (("miopen", "f32"), ("hip", "f32"), ("hip", "f16x3")):
    backbones.ENCODER_CONV, backbones.ENCODER_CONV_PRECISION = conv, prec
    with torch.no_grad():
        for _ in range(3):
            net(img)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            net(img)
        e1.record()
        torch.cuda.synchronize()
    print("ResnetFilter %-6s %-5s: %.3f ms/frame" % (conv, prec, e0.elapsed_time(e1) / 10))
