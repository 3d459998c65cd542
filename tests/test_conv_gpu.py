"""csrc/conv3x3.hip (fused GroupNorm -> ReLU -> conv3x3 on f32 MFMA) against the stock PyTorch ops
it replaces inside the encoders' pyramid blocks (backbones/HGFilters.py:40-62).  Needs an MI355X."""
import numpy as np
import pytest

from conftest import load_golden
from monoport_amd import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"

# (N, Cin, Cout, H, W): the shapes the hourglass encoder uses, each kernel instantiation
# (Cout 128 / 64 / 32) and tile shape (W = 128 / 64 / 32 / 256)
SHAPES = [(2, 256, 128, 128, 128), (1, 128, 64, 128, 128), (1, 64, 64, 128, 128), (3, 128, 64, 64, 64),
          (2, 256, 128, 32, 32), (1, 64, 32, 256, 256), (2, 32, 32, 64, 64), (1, 128, 128, 64, 64),
          (1, 64, 64, 32, 32), (1, 16, 32, 32, 32)]


def _ref_conv(x, gn, w):
    with torch.no_grad():
        v = torch.relu(gn(x)) if gn is not None else x
        # fp64 on the GPU: an order-independent reference for the 9 * Cin-term sums
        return torch.nn.functional.conv2d(v.double(), w.double(), padding=1)


@pytest.mark.parametrize("n,cin,cout,h,w", SHAPES)
def test_conv3x3_gn_matches_torch(n, cin, cout, h, w):
    from monoport_amd import ops
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + h)
    x = (torch.randn((n, cin, h, w), generator=g) * 2 + 0.3).to(DEV)
    wt = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(DEV)
    assert ops.conv3x3_supported(cin, cout, h, w)
    packed = ops.PackedConv3x3(wt)
    gn = None
    if cin % 32 == 0:
        gn = torch.nn.GroupNorm(32, cin).to(DEV)
        with torch.no_grad():
            gn.weight.uniform_(0.5, 1.5)
            gn.bias.uniform_(-0.5, 0.5)
        ss = ops.gn_finalize(ops.gn_stats(x, 32), n, cin, 32, (cin // 32) * h * w, gn.weight, gn.bias, gn.eps)
        # (scale, shift) against GroupNorm's definition
        xg = x.double().reshape(n, 32, -1)
        mean, var = xg.mean(2), xg.var(2, unbiased=False)
        rstd = 1.0 / torch.sqrt(var + gn.eps)
        sc = (rstd[:, :, None] * gn.weight.double().reshape(1, 32, -1)).reshape(n, cin)
        sh = gn.bias.double()[None] - (mean[:, :, None].expand(-1, -1, cin // 32).reshape(n, cin)) * sc
        assert (ss[..., 0].double() - sc).abs().max().item() <= 1e-5
        assert (ss[..., 1].double() - sh).abs().max().item() <= 1e-5
    else:
        ss = None
    y, stats = ops.conv3x3_gn(x, ss, packed, relu=gn is not None, want_stats=True)
    ref = _ref_conv(x, gn, wt)
    err = (y.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print("conv %s: max|d| %.3g (max|ref| %.3g)" % ((n, cin, cout, h, w), err, scale))
    assert y.shape == ref.shape and err <= 2e-5 * max(1.0, scale)
    # epilogue statistics -> the NEXT GroupNorm's scale / shift
    gn2 = torch.nn.GroupNorm(32, cout).to(DEV)
    with torch.no_grad():
        gn2.weight.uniform_(0.5, 1.5)
        gn2.bias.uniform_(-0.5, 0.5)
    ss2 = ops.gn_finalize(stats, n, cout, 32, (cout // 32) * h * w, gn2.weight, gn2.bias, gn2.eps)
    ss2_ref = ops.gn_finalize(ops.gn_stats(y, 32), n, cout, 32, (cout // 32) * h * w, gn2.weight, gn2.bias,
                              gn2.eps)
    assert (ss2 - ss2_ref).abs().max().item() <= 2e-5
    with torch.no_grad():
        want = torch.relu(gn2(y))
    got = torch.relu(y * ss2[..., 0, None, None] + ss2[..., 1, None, None])
    assert (got - want).abs().max().item() <= 5e-5


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("n,c,h,w", [(1, 256, 128, 128), (2, 64, 32, 64)])
def test_conv3x3_reflection_padding(precision, n, c, h, w):
    """reflect=True is nn.ReflectionPad2d(1) + an unpadded 3x3 convolution -- the residual blocks
    of the netC encoder (ResBlkFilters.py:28-84) -- with the GroupNorm + ReLU applied to the
    mirrored halo as well; followed by the block's x + GroupNorm(.) tail."""
    from monoport_amd import ops
    g = torch.Generator().manual_seed(c + h)
    x = (torch.randn((n, c, h, w), generator=g) * 2 + 0.3).to(DEV)
    wt = (torch.randn((c, c, 3, 3), generator=g) * (2.0 / (9 * c)) ** 0.5).to(DEV)
    gn = torch.nn.GroupNorm(32, c).to(DEV)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
    packed = ops.PackedConv3x3(wt, precision)
    pad = torch.nn.ReflectionPad2d(1)
    for ss_on in (False, True):
        ss = None
        if ss_on:
            ss = ops.gn_finalize(ops.gn_stats(x, 32), n, c, 32, (c // 32) * h * w, gn.weight, gn.bias, gn.eps)
        y, stats = ops.conv3x3_gn(x, ss, packed, relu=ss_on, want_stats=True, reflect=True)
        with torch.no_grad():
            v = torch.relu(gn(x)) if ss_on else x
            ref = torch.nn.functional.conv2d(pad(v).double(), wt.double())
        err = (y.double() - ref).abs().max().item()
        print("reflect conv %s %s gn=%d: max|d| %.3g" % (precision, (n, c, h, w), ss_on, err))
        assert err <= 2e-5 * max(1.0, ref.abs().max().item())
        # zero padding must differ on the border and agree in the interior
        y0, _ = ops.conv3x3_gn(x, ss, packed, relu=ss_on)
        assert torch.equal(y0[:, :, 1:-1, 1:-1], y[:, :, 1:-1, 1:-1]) and not torch.equal(y0, y)
    ss2 = ops.gn_finalize(stats, n, c, 32, (c // 32) * h * w, gn.weight, gn.bias, gn.eps)
    out = ops.scale_shift_add(y, ss2, x)
    with torch.no_grad():
        want = x + gn(y)
    assert (out - want).abs().max().item() <= 5e-5


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_resnet_filter_fused_equals_unfused(monkeypatch, precision):
    """The netC encoder with its twelve 256->256 reflect-padded convolutions on csrc/conv3x3.hip
    against the same module on the stock ops."""
    from monoport_amd.modeling import backbones
    net = backbones.ResnetFilter().eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(shapes, 5).items()})
    net.to(DEV)
    img = torch.from_numpy(syn.synthetic_image(6))[None].to(DEV)
    monkeypatch.setattr(backbones, "ENCODER_CONV_PRECISION", precision)
    with torch.no_grad():
        monkeypatch.setattr(backbones, "ENCODER_CONV", "miopen")
        ref = net(img)[0][0]
        monkeypatch.setattr(backbones, "ENCODER_CONV", "hip")
        assert net.model[-1]._fused_ok(ref)
        got = net(img)[0][0]
    err = (got - ref).abs().max().item()
    print("ResnetFilter fused (%s) vs stock ops: %.3g (max|ref| %.3g)" % (precision, err, ref.abs().max().item()))
    assert got.shape == (1, 256, 128, 128) and err <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("n,cin,cout,h,w", SHAPES[:8])
def test_conv3x3_f16x3_is_f32_class(n, cin, cout, h, w):
    """The split-f16 variant (three f16 MFMAs per product, f32 accumulation) against the fp64
    reference: held to the SAME bound as the exact-f32 kernel."""
    from monoport_amd import ops
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + h)
    x = (torch.randn((n, cin, h, w), generator=g) * 2 + 0.3).to(DEV)
    wt = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(DEV)
    gn = torch.nn.GroupNorm(32, cin).to(DEV) if cin % 32 == 0 else None
    ss = None
    if gn is not None:
        with torch.no_grad():
            gn.weight.uniform_(0.5, 1.5)
            gn.bias.uniform_(-0.5, 0.5)
        ss = ops.gn_finalize(ops.gn_stats(x, 32), n, cin, 32, (cin // 32) * h * w, gn.weight, gn.bias, gn.eps)
    y16, st16 = ops.conv3x3_gn(x, ss, ops.PackedConv3x3(wt, "f16x3"), relu=gn is not None, want_stats=True)
    y32, st32 = ops.conv3x3_gn(x, ss, ops.PackedConv3x3(wt, "f32"), relu=gn is not None, want_stats=True)
    ref = _ref_conv(x, gn, wt)
    e16 = (y16.double() - ref).abs().max().item()
    e32 = (y32.double() - ref).abs().max().item()
    print("conv f16x3 %s: |f16x3 - f64| %.3g, |f32 - f64| %.3g, |f16x3 - f32| %.3g"
          % ((n, cin, cout, h, w), e16, e32, (y16 - y32).abs().max().item()))
    assert e16 <= 2e-5 * max(1.0, ref.abs().max().item())
    # the two kernels may tile the launch differently (different slot counts): compare the sums per group
    assert (st16[0].sum(2) - st32[0].sum(2)).abs().max().item() <= 1e-2 * st32[1]


def test_encoder_f16x3_convs_vs_reference_golden(monkeypatch):
    """The whole hourglass encoder with every pyramid-block convolution on the split-f16 kernels
    (all map sizes) against the REFERENCE's CPU output: the same 1e-4 bar as fp32."""
    from monoport_amd.modeling import PIFuNetG, backbones
    monkeypatch.setattr(backbones, "ENCODER_CONV_PRECISION", "f16x3")
    g = load_golden("encoders")
    net = PIFuNetG().eval()
    shapes = {k: tuple(v.shape) for k, v in net.image_filter.state_dict().items()}
    net.image_filter.load_state_dict(
        {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(shapes, 71).items()})
    net.image_filter.to(DEV)
    img = torch.from_numpy(syn.synthetic_image(73))[None].to(DEV)
    with torch.no_grad():
        fg = net.filter(img)
    for i in range(4):
        err = float(np.abs(fg[i][0][0, ::8, ::8, ::8].cpu().numpy() - g["G%d" % i]).max())
        print("HGFilter stack %d on f16x3 convs vs reference: %.3g" % (i, err))
        assert err <= 1e-4


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_conv1x1_fused_tail_matches_torch(precision):
    """conv1x1_kernel: plain / GroupNorm-fused input, bias, second K segment + residual, the
    statistics epilogue and the channels-last output, against fp64 torch ops."""
    from monoport_amd import ops
    g = torch.Generator().manual_seed(11)
    n, h, w = 2, 64, 64
    y = (torch.randn((n, 256, h, w), generator=g) * 1.5).to(DEV)
    x = torch.randn((n, 256, h, w), generator=g).to(DEV)
    convs = [torch.nn.Conv2d(256, 256, 1).to(DEV) for _ in range(4)]  # conv_last, l, bl, al
    gn = torch.nn.GroupNorm(32, 256).to(DEV)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
        p_last = ops.PackedConv1x1(convs[0].weight, convs[0].bias, precision=precision)
        p_l = ops.PackedConv1x1(convs[1].weight, convs[1].bias, precision=precision)
        p_blal = ops.PackedConv1x1(convs[2].weight, convs[2].bias, convs[3].weight, convs[3].bias,
                                   precision=precision)
        t, st = ops.conv1x1(y, None, False, None, p_last, want_stats=True)
        t_ref = torch.nn.functional.conv2d(y.double(), convs[0].weight.double(), convs[0].bias.double())
        assert (t.double() - t_ref).abs().max().item() <= 2e-5 * max(1.0, t_ref.abs().max().item())
        ss = ops.gn_finalize(st, n, 256, 32, 8 * h * w, gn.weight, gn.bias, gn.eps)
        ss_ref = ops.gn_finalize(ops.gn_stats(t, 32), n, 256, 32, 8 * h * w, gn.weight, gn.bias, gn.eps)
        assert (ss - ss_ref).abs().max().item() <= 2e-5
        v = torch.relu(gn.double()(t_ref)) if False else torch.relu(
            torch.nn.functional.group_norm(t_ref, 32, gn.weight.double(), gn.bias.double(), gn.eps))
        hwc = torch.empty((n, h, w, 256), device=DEV)
        out, _ = ops.conv1x1(t, ss, True, None, p_l, y_hwc=hwc)
        out_ref = torch.nn.functional.conv2d(v, convs[1].weight.double(), convs[1].bias.double())
        e_out = (out.double() - out_ref).abs().max().item()
        assert e_out <= 5e-5 * max(1.0, out_ref.abs().max().item())
        assert torch.equal(hwc, out.permute(0, 2, 3, 1).contiguous())  # same values, channels-last
        only_hwc, _ = ops.conv1x1(t, ss, True, None, p_l, want_nchw=False, y_hwc=hwc)
        assert only_hwc is None
        xn, _ = ops.conv1x1(t, ss, True, out, p_blal, res=x)
        xn_ref = (x.double() + torch.nn.functional.conv2d(v, convs[2].weight.double(), convs[2].bias.double())
                  + torch.nn.functional.conv2d(out_ref, convs[3].weight.double(), convs[3].bias.double()))
        e_x = (xn.double() - xn_ref).abs().max().item()
        print("conv1x1 %s: |conv_last| ok, |l| %.3g, |x + bl + al| %.3g" % (precision, e_out, e_x))
        assert e_x <= 5e-5 * max(1.0, xn_ref.abs().max().item())


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("cin,cout,hw", [(64, 128, 128), (128, 256, 64)])
def test_conv1x1_projection_shortcut(precision, cin, cout, hw):
    """The 1x1 projection of a pyramid block whose channel count changes (HGFilters.py:47-52:
    GroupNorm -> ReLU -> Conv2d(Cin, Cout, 1, bias=False)), 128 or 256 output channels, no bias."""
    from monoport_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    n = 2
    x = (torch.randn((n, cin, hw, hw), generator=g) * 1.3 + 0.2).to(DEV)
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).to(DEV)
    gn = torch.nn.GroupNorm(32, cin).to(DEV)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
        packed = ops.PackedConv1x1(conv.weight, None, precision=precision)
        ss = ops.gn_finalize(ops.gn_stats(x, 32), n, cin, 32, (cin // 32) * hw * hw, gn.weight, gn.bias, gn.eps)
        y, st = ops.conv1x1(x, ss, True, None, packed)
        ref = torch.nn.functional.conv2d(torch.relu(gn(x)).double(), conv.weight.double())
        err = (y.double() - ref).abs().max().item()
        print("conv1x1 projection %s %d -> %d: %.3g" % (precision, cin, cout, err))
        assert st is None and y.shape == (n, cout, hw, hw) and err <= 2e-5 * max(1.0, ref.abs().max().item())
        if cout == 128:  # statistics / channels-last output exist for 256 rows only
            from monoport_amd._lib import MonoportError
            with pytest.raises(MonoportError):
                ops.conv1x1(x, ss, True, None, packed, want_stats=True)


def test_encoder_hwc_output_equals_packed_nchw():
    """HGFilter.forward(hwc_out=...): the last stack's features written channels-last by the
    producing kernel equal mp_feat_pack_hwc of the NCHW output, bit for bit."""
    from monoport_amd import ops
    from monoport_amd.modeling import PIFuNetG
    net = PIFuNetG().eval()
    shapes = {k: tuple(v.shape) for k, v in net.image_filter.state_dict().items()}
    net.image_filter.load_state_dict(
        {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(shapes, 71).items()})
    net.image_filter.to(DEV)
    img = torch.stack([torch.from_numpy(syn.synthetic_image(s)) for s in (73, 74)]).to(DEV)
    hwc = torch.empty((2, 128, 128, 256), device=DEV)
    with torch.no_grad():
        outs = net.image_filter(img, hwc_out=hwc)
        assert len(outs) == 4
        for b in range(2):
            assert torch.equal(hwc[b], ops.pack_features(outs[-1][0][b:b + 1]))
        only = net.image_filter(img, last_only=True, hwc_out=torch.empty_like(hwc))
        assert only[-1][0] is None


def test_conv3x3_rejects_unsupported_shapes():
    from monoport_amd import ops
    from monoport_amd._lib import MonoportError
    assert not ops.conv3x3_supported(3, 64, 512, 512)     # Cin % 16
    assert not ops.conv3x3_supported(64, 48, 64, 64)      # Cout % 32
    assert not ops.conv3x3_supported(64, 64, 24, 24)      # W not a power of two >= 32
    x = torch.zeros((1, 64, 24, 24), device=DEV)
    packed = ops.PackedConv3x3(torch.zeros((64, 64, 3, 3), device=DEV))
    with pytest.raises(MonoportError):
        ops.conv3x3_gn(x, None, packed)


def test_convblock_fused_equals_unfused_module(monkeypatch):
    """The pyramid block on the fused kernels vs the same module on MIOpen convolutions + the
    stand-alone GroupNorm kernel."""
    from monoport_amd.modeling import backbones
    for c_in, c_out, hw, n in ((256, 256, 128, 2), (128, 256, 64, 1), (64, 128, 256, 1), (256, 256, 32, 3)):
        blk = backbones.ConvBlock(c_in, c_out)
        shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
        blk.load_state_dict({k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(shapes, 5).items()})
        blk.to(DEV).eval()
        x = torch.randn((n, c_in, hw, hw), generator=torch.Generator().manual_seed(c_in + hw)).to(DEV)
        with torch.no_grad():
            monkeypatch.setattr(backbones, "ENCODER_CONV", "hip")
            monkeypatch.setattr(backbones, "ENCODER_CONV_MIN_H", 32)
            assert blk._fused_ok(x)
            fused = blk(x)
            monkeypatch.setattr(backbones, "ENCODER_CONV", "miopen")
            assert not blk._fused_ok(x)
            plain = blk(x)
        err = (fused - plain).abs().max().item()
        print("ConvBlock(%d,%d)@%d: |fused - miopen| %.3g (max %.3g)" % (c_in, c_out, hw, err, plain.abs().max().item()))
        assert err <= 1e-4 * max(1.0, plain.abs().max().item())


def test_encoder_with_fused_convs_vs_reference_golden():
    """The whole hourglass encoder on the fused kernels against the REFERENCE's CPU run of the
    same seeded weights (fixture G0-G3), and against the MIOpen path."""
    from monoport_amd.modeling import PIFuNetG, backbones
    assert backbones.ENCODER_CONV == "hip"
    g = load_golden("encoders")
    net = PIFuNetG().eval()
    shapes = {k: tuple(v.shape) for k, v in net.image_filter.state_dict().items()}
    net.image_filter.load_state_dict(
        {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(shapes, 71).items()})
    net.image_filter.to(DEV)
    img = torch.from_numpy(syn.synthetic_image(73))[None].to(DEV)
    with torch.no_grad():
        fg = net.filter(img)
    for i in range(4):
        err = float(np.abs(fg[i][0][0, ::8, ::8, ::8].cpu().numpy() - g["G%d" % i]).max())
        print("HGFilter stack %d on fused convs vs reference: %.3g" % (i, err))
        assert err <= 1e-4
