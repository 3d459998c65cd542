"""Constants and seeded fixtures shared by bench.py, bench_dropin.py and the probes under tools/ (round 6: split out
of bench.py)."""
import torch

from monoport_amd import synthetic as syn
from monoport_amd.modeling import PIFuNetC, PIFuNetG

RESOLUTIONS = [17, 33, 65, 129, 257]  # RTL/main.py:187
B_MIN, B_MAX = [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]  # RTL/main.py:185-186
FLOP_PER_POINT = 2363906  # netG MLP, SURVEY.md section 8d / BASELINE.md section 2
# with the skip tables (mp_skip_table, default): the products of weights with the sampled feature
# (layer 0 and the skip connections: 1921 x 256 multiply-adds) leave the per-point work -- they are
# taken once per texel and frame in skip_table_kernel (16 GFLOP per frame)
FLOP_PER_POINT_SKIP_TABLE = FLOP_PER_POINT - 2 * 1921 * 256
FLOP_SKIP_TABLE_PER_FRAME = 2 * 1921 * 256 * 128 * 128
FLOP_PER_POINT_C = 3350022  # netC MLP (per-vertex colour query)
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E spec peak
N_IMAGES = 8  # distinct synthetic frames cycled through the timed region
# the reference's OWN modules timed on CPU (oracle/time_reference.py, build container: the GPU box
# has no /root/reference) -- a constant with its provenance, next to the live "port" baseline
CPU_BASELINE_REFERENCE = {
    "value": 0.123, "unit": "recon/s", "cores": 8, "kind": "reference",
    "where": "build container (no GPU), 8 threads; not re-measured on the GPU box",
    "source": "oracle/time_reference.py -> BASELINE.md section 4",
    "sample": "1 reconstruction = netG.filter 0.565 s + 17..257 octree through the reference's "
              "netG.query 7.49 s (280,936 points) + forward_vertices 0.092 s = 8.15 s",
}


def set_precision_everywhere(head, precision):
    """MLP arithmetic of the fused query kernel AND of the encoder's fused 3x3 convolutions:
    "f16x3" switches both to f32 emulated on f16 MFMA (three MFMAs per product, f32 accumulate);
    the other f16 query variants leave the encoder on exact f32."""
    from monoport_amd.modeling import backbones
    head.set_precision(precision)
    backbones.set_encoder_conv_precision("f16x3" if precision == "f16x3" else "f32")


def build_netg(device, precision="f32"):
    """Random-init (seeded) encoder of the reference architecture + the analytic F-body head."""
    net = PIFuNetG().eval()
    set_precision_everywhere(net.surface_classifier, precision)
    shapes = {k: tuple(v.shape) for k, v in net.image_filter.state_dict().items()}
    sd = syn.seeded_state_dict(shapes, 71)
    net.image_filter.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    layers = syn.body_mlp("G", noise=0.05, seed=1)
    net.surface_classifier.load_state_dict(
        {**{"filters.%d.weight" % i: torch.from_numpy(w)[:, :, None] for i, (w, _) in enumerate(layers)},
         **{"filters.%d.bias" % i: torch.from_numpy(b) for i, (_, b) in enumerate(layers)}})
    return net.to(device), layers


def build_netc(device):
    """netC with seeded random weights of the reference architecture (config 3)."""
    net = PIFuNetC().eval()
    shapes = {k: tuple(v.shape) for k, v in net.image_filter.state_dict().items()}
    sd = syn.seeded_state_dict(shapes, 72)
    net.image_filter.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    layers = syn.rand_mlp("C", 61, 2.0)
    net.surface_classifier.load_state_dict(
        {**{"filters.%d.weight" % i: torch.from_numpy(w)[:, :, None] for i, (w, _) in enumerate(layers)},
         **{"filters.%d.bias" % i: torch.from_numpy(b) for i, (_, b) in enumerate(layers)}})
    return net.to(device)
