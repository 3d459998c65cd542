"""Round 3: GroupNorm hand-over inside the producing kernels (csrc/gn_tail.h), the pyramid block's
cat + residual written by the convolution epilogues, the split-K 3x3 kernel for small launches, the
im2col kernel for the stems / stride-2 convolutions (csrc/convim2col.hip) and the encoders built from
them -- against stock PyTorch ops in fp64 and the REFERENCE's encoder goldens.  Needs an MI355X."""
import numpy as np
import pytest

from conftest import check_full_coverage, load_golden
from monoport_amd import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"


def _gn(c, seed):
    g = torch.Generator().manual_seed(seed)
    gn = torch.nn.GroupNorm(32, c)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        gn.bias.copy_(torch.rand(c, generator=g) - 0.5)
    return gn.to(DEV)


def _ss_ref(t, gn):
    """(scale, shift) of GroupNorm gn over t [N,C,H,W], from the definition, in fp64."""
    n, c = t.shape[0], t.shape[1]
    tg = t.double().reshape(n, 32, -1)
    mean, var = tg.mean(2), tg.var(2, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + gn.eps)
    cpg = c // 32
    sc = (rstd[:, :, None] * gn.weight.double().reshape(1, 32, cpg)).reshape(n, c)
    sh = gn.bias.double()[None] - mean[:, :, None].expand(-1, -1, cpg).reshape(n, c) * sc
    return torch.stack((sc, sh), 2)


def _check_acc(acc, t, gn, what, channels=None, tol=2e-5):
    """The (scale, shift) a consumer derives from accumulator ``acc`` against GroupNorm's definition
    over t (``channels``: only this slice was accumulated)."""
    from monoport_amd import ops
    c = t.shape[1]
    count = (c // 32) * t.shape[2] * t.shape[3]
    got = ops.gn_reference_ss(acc, gn, count).double()
    ref = _ss_ref(t, gn)
    if channels is not None:
        got, ref = got[:, channels], ref[:, channels]
    err = (got - ref).abs().max().item()
    assert got.shape == ref.shape and err <= tol * max(1.0, ref.abs().max().item()), "%s: ss off by %g" % (what, err)


def _acc(n):
    from monoport_amd import ops
    return ops.gn_acc_zeros(DEV, n)


def _acc_of(x):
    """Accumulator holding the statistics of x (through mp_gn_apply with an identity scale / shift)."""
    from monoport_amd import ops
    ident = torch.zeros((x.shape[0], x.shape[1], 2), device=DEV)
    ident[..., 0] = 1.0
    acc = _acc(x.shape[0])
    y = ops.gn_apply(x, ident, relu=False, stats=acc)
    assert torch.equal(y, x)
    return acc


# (N, Cin, Cout, H, W, Ctot, off): Ctot / off = the pyramid block this convolution fills
CASES = [(1, 256, 128, 128, 128, 256, 0), (1, 128, 64, 128, 128, 256, 128), (1, 64, 64, 128, 128, 256, 192),
         (1, 256, 128, 64, 64, 256, 0), (1, 128, 64, 32, 32, 256, 128), (2, 64, 64, 32, 32, 256, 192),
         (1, 64, 32, 256, 256, 128, 64), (1, 32, 32, 256, 256, 128, 96), (3, 256, 128, 32, 32, 256, 0),
         (10, 128, 64, 64, 64, 256, 128), (1, 16, 32, 32, 32, 128, 32)]


@pytest.mark.parametrize("mode", ["auto", "large", "splitk"])
@pytest.mark.parametrize("n,cin,cout,h,w,ctot,off", CASES)
def test_conv3x3_fused_handover_and_tail(mode, n, cin, cout, h, w, ctot, off):
    from monoport_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin * 7 + cout + h + n)
    x = (torch.randn((n, cin, h, w), generator=g) * 2 + 0.3).to(DEV)
    res = torch.randn((n, ctot, h, w), generator=g).to(DEV)
    wt = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(DEV)
    gn_in = _gn(cin, 1) if cin % 32 == 0 else None
    gn_y, gn_o = _gn(cout, 2), _gn(ctot, 3)
    acc_x = _acc_of(x) if gn_in is not None else None
    packed = ops.PackedConv3x3(wt)
    out = torch.full((n, ctot, h, w), 7.0, device=DEV)
    acc_y, acc_o, acc_y2, acc_o2 = _acc(n), _acc(n), _acc(n), _acc(n)
    gn_arg = (acc_x, gn_in) if gn_in is not None else None
    lib.mp_conv3x3_tune({"auto": 0, "large": 0x100, "splitk": 0x200}[mode])
    try:
        y = ops.conv3x3_fused(x, gn_arg, packed, relu=gn_in is not None, stats=acc_y, out=out, res=res,
                              out_off=off, out_stats=acc_o)
        y2 = ops.conv3x3_fused(x, gn_arg, packed, relu=gn_in is not None, stats=acc_y2, out=out.clone(), res=res,
                               out_off=off, out_stats=acc_o2)
        if gn_in is not None:  # the legacy form of the same input GroupNorm: precomputed (scale, shift)
            ss_in = ops.gn_reference_ss(acc_x, gn_in, (cin // 32) * h * w)
            y3 = ops.conv3x3_fused(x, ss_in, packed, relu=True)
    finally:
        lib.mp_conv3x3_tune(0)
    with torch.no_grad():
        v = x.double()
        if gn_in is not None:
            ss64 = _ss_ref(x, gn_in)
            v = torch.relu(v * ss64[..., 0, None, None] + ss64[..., 1, None, None])
        ref = torch.nn.functional.conv2d(v, wt.double(), padding=1)
    err = (y.double() - ref).abs().max().item()
    print("conv3x3_fused %s %s: max|d| %.3g" % (mode, (n, cin, cout, h, w), err))
    assert err <= 3e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(y, y2) and torch.equal(acc_y, acc_y2) and torch.equal(acc_o, acc_o2)  # deterministic
    if gn_in is not None:
        assert torch.equal(y, y3)  # hand-over and precomputed (scale, shift) agree bit for bit
    _check_acc(acc_y, y, gn_y, "raw output")
    # the block tail: this launch's channels of cat + residual, everything else untouched
    assert torch.equal(out[:, off:off + cout], y + res[:, off:off + cout])
    untouched = torch.ones(ctot, dtype=torch.bool)
    untouched[off:off + cout] = False
    assert (out[:, untouched] == 7.0).all()
    _check_acc(acc_o, out, gn_o, "block output", channels=slice(off, off + cout))
    cpg = ctot // 32
    assert (acc_o[:, :, :off // cpg] == 0).all() and (acc_o[:, :, (off + cout) // cpg:] == 0).all()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_conv3x3_fused_f16x3_and_reflect(precision):
    from monoport_amd import ops
    n, c, h, w = 2, 256, 64, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn((n, c, h, w), generator=g).to(DEV)
    wt = (torch.randn((c, c, 3, 3), generator=g) * (2.0 / (9 * c)) ** 0.5).to(DEV)
    gn = _gn(c, 8)
    acc = _acc(n)
    y = ops.conv3x3_fused(x, None, ops.PackedConv3x3(wt, precision), relu=False, reflect=True, stats=acc)
    ref = torch.nn.functional.conv2d(torch.nn.ReflectionPad2d(1)(x).double(), wt.double())
    assert (y.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    _check_acc(acc, y, gn, "reflect conv")
    # the block's second half: GroupNorm + ReLU handed over, then x + GroupNorm(.) (ResBlkFilters.py:75-84)
    gn2 = _gn(c, 9)
    acc2 = _acc(n)
    u = ops.conv3x3_fused(y, (acc, gn), ops.PackedConv3x3(wt, precision), relu=True, reflect=True, stats=acc2)
    out = ops.gn_apply(u, (acc2, gn2), False, res=x)
    with torch.no_grad():
        v = torch.relu(gn(y))
        u_ref = torch.nn.functional.conv2d(torch.nn.ReflectionPad2d(1)(v).double(), wt.double())
        want = x + gn2(u)
    assert (u.double() - u_ref).abs().max().item() <= 5e-5 * max(1.0, u_ref.abs().max().item())
    assert (out - want).abs().max().item() <= 5e-5


@pytest.mark.parametrize("mrw", [0, 1, 2])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_conv1x1_fused_handover(precision, mrw):
    from monoport_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(11 + mrw)
    n, h, w = 2, 64, 64
    y = (torch.randn((n, 256, h, w), generator=g) * 1.5).to(DEV)
    x = torch.randn((n, 256, h, w), generator=g).to(DEV)
    convs = [torch.nn.Conv2d(256, 256, 1).to(DEV) for _ in range(4)]  # conv_last, l, bl, al
    gn_end, gn_next = _gn(256, 21), _gn(256, 22)
    lib.mp_conv3x3_tune(mrw << 12)
    try:
        with torch.no_grad():
            p_last = ops.PackedConv1x1(convs[0].weight, convs[0].bias, precision=precision)
            p_l = ops.PackedConv1x1(convs[1].weight, convs[1].bias, precision=precision)
            p_blal = ops.PackedConv1x1(convs[2].weight, convs[2].bias, convs[3].weight, convs[3].bias,
                                       precision=precision)
            acc_t, acc_x = _acc(n), _acc(n)
            t = ops.conv1x1_fused(y, None, False, None, p_last, stats=acc_t)
            t_ref = torch.nn.functional.conv2d(y.double(), convs[0].weight.double(), convs[0].bias.double())
            assert (t.double() - t_ref).abs().max().item() <= 2e-5 * max(1.0, t_ref.abs().max().item())
            _check_acc(acc_t, t, gn_end, "conv_last")
            hwc = torch.empty((n, h, w, 256), device=DEV)
            out = ops.conv1x1_fused(t, (acc_t, gn_end), True, None, p_l, y_hwc=hwc)
            v = torch.relu(gn_end(t)).double()
            out_ref = torch.nn.functional.conv2d(v, convs[1].weight.double(), convs[1].bias.double())
            assert (out.double() - out_ref).abs().max().item() <= 5e-5 * max(1.0, out_ref.abs().max().item())
            assert torch.equal(hwc, out.permute(0, 2, 3, 1).contiguous())
            xn = ops.conv1x1_fused(t, (acc_t, gn_end), True, out, p_blal, res=x, stats=acc_x)
            xn_ref = (x.double() + torch.nn.functional.conv2d(v, convs[2].weight.double(), convs[2].bias.double())
                      + torch.nn.functional.conv2d(out.double(), convs[3].weight.double(), convs[3].bias.double()))
            assert (xn.double() - xn_ref).abs().max().item() <= 5e-5 * max(1.0, xn_ref.abs().max().item())
            _check_acc(acc_x, xn, gn_next, "x + bl + al")
    finally:
        lib.mp_conv3x3_tune(0)


def test_convk_stem_and_downsampling():
    from monoport_amd import ops
    g = torch.Generator().manual_seed(3)
    img = torch.from_numpy(np.stack([syn.synthetic_image(s) for s in (3, 4)])).to(DEV)
    # hourglass stem: 7x7 stride 2, zero padding 3, bias
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3).to(DEV)
    gn = _gn(64, 31)
    acc = _acc(2)
    with torch.no_grad():
        y = ops.convk(img, None, False, ops.PackedConvK(conv.weight, conv.bias), 2, stats=acc)
        ref = torch.nn.functional.conv2d(img.double(), conv.weight.double(), conv.bias.double(), stride=2, padding=3)
    e = (y.double() - ref).abs().max().item()
    print("stem 7x7 s2: %.3g" % e)
    assert y.shape == (2, 64, 256, 256) and e <= 2e-5 * max(1.0, ref.abs().max().item())
    _check_acc(acc, y, gn, "stem")
    # netC stem: ReflectionPad2d(3) + 7x7, no bias
    conv7 = torch.nn.Conv2d(3, 64, 7, bias=False).to(DEV)
    gn7 = _gn(64, 32)
    acc7 = _acc(2)
    with torch.no_grad():
        t = ops.convk(img, None, False, ops.PackedConvK(conv7.weight), 1, reflect=True, stats=acc7)
        ref7 = torch.nn.functional.conv2d(torch.nn.ReflectionPad2d(3)(img).double(), conv7.weight.double())
    e = (t.double() - ref7).abs().max().item()
    print("netC stem 7x7 reflect: %.3g" % e)
    assert t.shape == (2, 64, 512, 512) and e <= 2e-5 * max(1.0, ref7.abs().max().item())
    _check_acc(acc7, t, gn7, "netC stem")
    # stride-2 3x3 with the previous GroupNorm + ReLU applied while gathering
    for cin, cout, src, gnp, acc_in in ((64, 128, t, gn7, acc7), (128, 256, None, None, None)):
        if src is None:
            src = (torch.randn((2, cin, 256, 256), generator=g) * 1.3).to(DEV)
            gnp = _gn(cin, 33)
            acc_in = _acc_of(src)
        cv = torch.nn.Conv2d(cin, cout, 3, 2, 1, bias=False).to(DEV)
        gno = _gn(cout, 34)
        accd = _acc(2)
        with torch.no_grad():
            d = ops.convk(src, (acc_in, gnp), True, ops.PackedConvK(cv.weight), 2, stats=accd)
            refd = torch.nn.functional.conv2d(torch.relu(gnp(src)).double(), cv.weight.double(), stride=2, padding=1)
        e = (d.double() - refd).abs().max().item()
        print("3x3 s2 %d -> %d: %.3g" % (cin, cout, e))
        assert d.shape == refd.shape and e <= 5e-5 * max(1.0, refd.abs().max().item())
        _check_acc(accd, d, gno, "3x3 s2")


def test_elementwise_producers_with_handover():
    from monoport_amd import ops
    g = torch.Generator().manual_seed(9)
    for n, c, h in ((1, 256, 128), (3, 128, 32), (2, 64, 256)):
        x = (torch.randn((n, c, h, h), generator=g) * 1.7 + 0.2).to(DEV)
        gn_a, gn_b = _gn(c, 41), _gn(c, 42)
        acc = _acc(n)
        y = ops.avgpool2_gn(x, acc)
        ref = torch.nn.functional.avg_pool2d(x, 2, stride=2)
        assert (y - ref).abs().max().item() <= 1e-6
        _check_acc(acc, y, gn_a, "avgpool")
        if h <= 128:
            skip = torch.randn((n, c, 2 * h, 2 * h), generator=g).to(DEV)
            acc_u = _acc(n)
            u = ops.upsample_add_gn(x, skip, acc_u)
            assert torch.equal(u, ops.upsample_bicubic2x(x, add=skip))
            want = skip + torch.nn.functional.interpolate(x, scale_factor=2, mode="bicubic", align_corners=True)
            assert (u - want).abs().max().item() <= 1e-4
            _check_acc(acc_u, u, gn_a, "upsample_add a")
            _check_acc(acc_u, u, gn_b, "upsample_add b")  # two readers, one accumulator
        acc_x, acc_z = _acc_of(x), _acc(n)
        z = ops.gn_apply(x, (acc_x, gn_b), True, stats=acc_z)
        with torch.no_grad():
            want = torch.relu(gn_b(x))
        assert (z - want).abs().max().item() <= 5e-5
        _check_acc(acc_z, z, gn_a, "gn_apply")
        z2 = ops.gn_apply(x, (acc_x, gn_b), False)
        with torch.no_grad():
            assert (z2 - gn_b(x)).abs().max().item() <= 5e-5


def _netg(seed=71):
    from monoport_amd.modeling import PIFuNetG
    net = PIFuNetG().eval()
    shapes = {k: tuple(v.shape) for k, v in net.image_filter.state_dict().items()}
    net.image_filter.load_state_dict(
        {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(shapes, seed).items()})
    net.image_filter.to(DEV)
    return net


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_hgfilter_dataflow_vs_reference_and_round2_path(monkeypatch, precision):
    """The hourglass encoder as a chain of hand-written kernels only (stem included) against the
    REFERENCE's CPU output (1e-4) and against round 2's per-module path; batch 1 and batch 3 (the
    small maps take the split-K kernel at batch 1, large tiles at batch 3)."""
    from monoport_amd.modeling import backbones
    monkeypatch.setattr(backbones, "ENCODER_CONV_PRECISION", precision)
    gold = load_golden("encoders")
    net = _netg()
    imgs = torch.stack([torch.from_numpy(syn.synthetic_image(s)) for s in (73, 74, 75)]).to(DEV)
    with torch.no_grad():
        assert net.image_filter._dataflow_ok(imgs)
        one = net.image_filter(imgs[:1])
        three = net.image_filter(imgs)
        again = net.image_filter(imgs)
        monkeypatch.setattr(backbones, "ENCODER_DATAFLOW", "off")
        assert not net.image_filter._dataflow_ok(imgs)
        old = net.image_filter(imgs)
    for i in range(4):
        err = float(np.abs(one[i][0][0, ::8, ::8, ::8].cpu().numpy() - gold["G%d" % i]).max())
        e3 = float(np.abs(three[i][0][0, ::8, ::8, ::8].cpu().numpy() - gold["G%d" % i]).max())
        d_old = (three[i][0] - old[i][0]).abs().max().item()
        d_b = (three[i][0][:1] - one[i][0]).abs().max().item()
        print("HGFilter dataflow %s stack %d: vs reference %.3g (batch 3: %.3g), vs round-2 path %.3g, "
              "batch 1 vs 3 %.3g" % (precision, i, err, e3, d_old, d_b))
        assert err <= 1e-4 and e3 <= 1e-4 and d_old <= 1e-4 and d_b <= 1e-4
        assert torch.equal(three[i][0], again[i][0])  # deterministic
    eb, ep = check_full_coverage(gold, "G3", one[3][0][0].cpu().numpy())
    print("HGFilter dataflow %s: full-coverage G3 block-mean error %.3g, pixel-mean error %.3g" % (precision, eb, ep))


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_folded_tail_of_unused_stacks(monkeypatch, precision):
    """last_only (what FramePipeline asks for: MonoPortNet.query keeps feats_stages[-1] only): stacks 0-2 hand
    over with ONE folded 1x1 GEMM, x + (W_bl + W_al W_l) y + b, instead of l and [bl | al] (HGFilters.py:187-204;
    HGFilter._tail_packed).  The last stack's features against the REFERENCE's (every element, through the
    8 x 8 block / pixel means of the fixture), against the three-GEMM form (another association of the same sums:
    f32 rounding only), batch 1 and 3, channels-last output included; MONOPORT_ENCODER_FOLD_TAIL=off = the
    three-GEMM form bit for bit."""
    from monoport_amd.modeling import backbones
    monkeypatch.setattr(backbones, "ENCODER_CONV_PRECISION", precision)
    assert backbones.ENCODER_FOLD_TAIL == "on"
    gold = load_golden("encoders")
    net = _netg()
    enc = net.image_filter
    imgs = torch.stack([torch.from_numpy(syn.synthetic_image(s)) for s in (73, 74, 75)]).to(DEV)
    with torch.no_grad():
        all3 = enc(imgs, graphed=False)            # all four outputs asked for: l computed, hand-over folded
        fold1, fold3 = enc(imgs[:1], last_only=True, graphed=False), enc(imgs, last_only=True, graphed=False)
        hwc = torch.empty((3, 128, 128, 256), device=DEV)
        only = enc(imgs, last_only=True, hwc_out=hwc, graphed=False)
        monkeypatch.setattr(backbones, "ENCODER_FOLD_TAIL", "off")
        full1, full3 = enc(imgs[:1], graphed=False), enc(imgs, graphed=False)      # the reference's three GEMMs
        plain3 = enc(imgs, last_only=True, graphed=False)
    assert len(fold1) == len(fold3) == 1 and only[-1][0] is None
    assert torch.equal(plain3[-1][0], full3[3][0])
    assert torch.equal(all3[3][0], fold3[-1][0]) and torch.equal(all3[0][0], full3[0][0])  # same hand-over; stack 0's l(y) untouched
    for i in range(1, 4):
        assert 0 < (all3[i][0] - full3[i][0]).abs().max().item() <= 2e-5
    d1 = (fold1[-1][0] - full1[3][0]).abs().max().item()
    d3 = (fold3[-1][0] - full3[3][0]).abs().max().item()
    assert torch.equal(hwc, fold3[-1][0].permute(0, 2, 3, 1))
    err = float(np.abs(fold1[-1][0][0, ::8, ::8, ::8].cpu().numpy() - gold["G3"]).max())
    eb, ep = check_full_coverage(gold, "G3", fold1[-1][0][0].cpu().numpy())
    print("folded tail %s: vs the three-GEMM form %.3g (batch 1) / %.3g (batch 3); vs the reference's G3 %.3g, "
          "block means %.3g, pixel means %.3g" % (precision, d1, d3, err, eb, ep))
    assert 0 < d1 <= 2e-5 and 0 < d3 <= 2e-5 and err <= 1e-4


def test_hgfilter_dataflow_hwc_and_last_only():
    from monoport_amd import ops
    net = _netg()
    img = torch.stack([torch.from_numpy(syn.synthetic_image(s)) for s in (73, 74)]).to(DEV)
    hwc = torch.empty((2, 128, 128, 256), device=DEV)
    with torch.no_grad():
        outs = net.image_filter(img, hwc_out=hwc)
        assert len(outs) == 4
        for b in range(2):
            assert torch.equal(hwc[b], ops.pack_features(outs[-1][0][b:b + 1]))
        only = net.image_filter(img, last_only=True, hwc_out=torch.empty_like(hwc))
        assert len(only) == 1 and only[-1][0] is None


def test_encoder_graph_replay_equals_eager():
    """graphed=True: the kernel chain replayed as one hipGraph gives the eager results bit for bit,
    returns tensors that survive the next call, and is re-captured when the weights change."""
    net = _netg()
    img = torch.stack([torch.from_numpy(syn.synthetic_image(s)) for s in (73, 74)]).to(DEV)
    enc = net.image_filter
    with torch.no_grad():
        eager = enc(img[:1], graphed=False)
        first = enc(img[:1], graphed=True)
        keep = first[3][0].clone()
        second = enc(img[1:], graphed=True)          # same graph, other image
        assert all(torch.equal(a[0], b[0]) for a, b in zip(eager, first))
        assert torch.equal(first[3][0], keep)         # not overwritten by the replay
        assert not torch.equal(second[3][0], keep)
        hwc = torch.empty((1, 128, 128, 256), device=DEV)
        only = enc(img[:1], last_only=True, hwc_out=hwc, graphed=True)
        eager_last = enc(img[:1], last_only=True, graphed=False)  # last_only folds the tails of stacks 0-2: its own bits
        assert only[-1][0] is None and torch.equal(hwc[0], eager_last[-1][0][0].permute(1, 2, 0))
        enc.conv1.bias.add_(0.25)                     # in-place update: the fingerprint changes
        changed = enc(img[:1], graphed=True)
        assert torch.equal(changed[3][0], enc(img[:1], graphed=False)[3][0])
        assert not torch.equal(changed[3][0], keep)


@pytest.mark.parametrize("pool", ["on", "off"])
@pytest.mark.parametrize("batch", [1, 2])
def test_encoder_plan_replay_equals_eager(batch, pool, monkeypatch):
    """The default of a drop-in netG.filter call at batch <= 2: its ~137 launches recorded once into an mp_plan
    (csrc/plan.hip; at these batches the hourglass's skip branches run on side streams, joined by events) and
    replayed by ONE C-ABI call.  Same bits as launch by launch, on the recording call and on every replay; the
    returned tensors survive later calls; other images give other results; a weight update records a new plan;
    MONOPORT_ENCODER_PLAN=off / larger batches run launch by launch."""
    from monoport_amd.modeling import backbones
    assert backbones.ENCODER_PLAN == "auto" and backbones.ENCODER_PLAN_MAX_BATCH >= 2 and backbones.ENCODER_PLAN_POOL == "on"
    monkeypatch.setattr(backbones, "ENCODER_PLAN", "on")  # "auto" = only inside per-frame StagePipeline stage threads (below)
    monkeypatch.setattr(backbones, "ENCODER_PLAN_POOL", pool)  # "on": intermediates recycled inside a private pool
    net = _netg()
    imgs = [torch.stack([torch.from_numpy(syn.synthetic_image(s + b)) for b in range(batch)]).to(DEV) for s in (73, 83, 93)]
    enc = net.image_filter
    with torch.no_grad():
        eager = [enc(im, graphed=False) for im in imgs]
        first = enc(imgs[0])                       # records
        plans = enc.__dict__["_plans"].entries
        assert len(plans) == 1
        plan = next(iter(plans.values()))[0]
        assert plan.n_cmds > 130
        keep = [o[0].clone() for o in first]
        # memory that is allocated, written and freed between replays must never be the plan's
        junk = [torch.full((1, 256, 128, 128), float(i), device=DEV) for i in range(24)]
        del junk
        second = enc(imgs[1])                      # replays on another image
        third = enc(imgs[2])
        again = enc(imgs[0])
        assert len(plans) == 1
        for got, ref in ((first, eager[0]), (second, eager[1]), (third, eager[2]), (again, eager[0])):
            assert len(got) == 4 and all(torch.equal(a[0], b[0]) for a, b in zip(got, ref))
        assert all(torch.equal(o[0], k) for o, k in zip(first, keep))   # outputs are not the plan's buffers
        assert not torch.equal(second[3][0], keep[3])
        # the channels-last output and last_only through a plan of their own
        hwc = torch.empty((batch, 128, 128, 256), device=DEV)
        only = enc(imgs[1], last_only=True, hwc_out=hwc)
        only2 = enc(imgs[2], last_only=True, hwc_out=hwc)
        assert only[-1][0] is None and only2[-1][0] is None and len(plans) == 2
        eager_last = enc(imgs[2], last_only=True, graphed=False)  # last_only folds the tails of stacks 0-2: its own bits
        assert torch.equal(hwc, eager_last[-1][0].permute(0, 2, 3, 1))
        # the whole module API: netG.filter -> list of 4 stages
        feats = net.filter(imgs[1])
        assert len(feats) == 4 and torch.equal(feats[-1][0], eager[1][3][0])
        enc.conv1.bias.add_(0.25)                  # in-place update: the fingerprint changes -> a new plan
        changed = enc(imgs[0])
        assert torch.equal(changed[3][0], enc(imgs[0], graphed=False)[3][0]) and not torch.equal(changed[3][0], keep[3])
        monkeypatch.setattr(backbones, "ENCODER_PLAN", "off")
        n = len(plans)
        off = enc(imgs[0])
        assert len(plans) == n and torch.equal(off[3][0], changed[3][0])
        monkeypatch.setattr(backbones, "ENCODER_PLAN", "on")
        big = torch.cat([imgs[0], imgs[1], imgs[2]])[:3]
        enc(big)                                   # batch 3 > ENCODER_PLAN_MAX_BATCH: no plan
        assert len(plans) == n
        # the default policy: a plan in a per-frame stage thread of a StagePipeline, none in the calling thread,
        # none in a coalescing stage
        monkeypatch.setattr(backbones, "ENCODER_PLAN", "auto")
        from monoport_amd.stage_pipeline import Coalesced, StagePipeline
        enc(imgs[0])
        assert len(plans) == n
        seen = []

        def stage(im):
            out = enc(im)
            seen.append(len(plans))
            return out[3][0]

        got = list(StagePipeline([imgs[0], imgs[1]], [stage], device=DEV, max_in_flight=1))
        assert seen[-1] == n + 1 and torch.equal(got[1], enc(imgs[1], graphed=False)[3][0])
        got = list(StagePipeline([imgs[0], imgs[1]], [Coalesced(stage, lambda ims: [stage(im) for im in ims])], device=DEV,
                                 max_in_flight=1))
        assert seen[-1] == n + 1 and torch.equal(got[0], enc(imgs[0], graphed=False)[3][0])


def test_plan_api_rejects_bad_commands():
    """mp_plan_add checks the size of every command's argument block and its stream slot."""
    import ctypes
    from monoport_amd import _lib, ops
    ctx = ops.get_context(DEV)
    handle = ctypes.c_void_p()
    ctx.check(ctx.lib.mp_plan_create(ctx.handle, 1, ctypes.byref(handle)), "mp_plan_create")
    try:
        m = _lib.PlanMemsetArgs()
        buf = torch.zeros(16, device=DEV)
        m.ptr, m.bytes, m.value = buf.data_ptr(), 64, 0
        blob = ctypes.create_string_buffer(bytes(m), ctypes.sizeof(m))
        assert ctx.lib.mp_plan_add(handle, _lib.PLAN_MEMSET, ctypes.cast(blob, ctypes.c_void_p), ctypes.sizeof(m), 0) == 0
        assert ctx.lib.mp_plan_add(handle, _lib.PLAN_MEMSET, ctypes.cast(blob, ctypes.c_void_p), 8, 0) != 0      # wrong size
        assert ctx.lib.mp_plan_add(handle, _lib.PLAN_MEMSET, ctypes.cast(blob, ctypes.c_void_p), ctypes.sizeof(m), 5) != 0  # no such slot
        assert ctx.lib.mp_plan_add(handle, 99, ctypes.cast(blob, ctypes.c_void_p), ctypes.sizeof(m), 0) != 0     # no such command
        w = _lib.PlanWaitArgs()
        w.waiter_slot, w.signaller_slot = 1, 1
        wb = ctypes.create_string_buffer(bytes(w), ctypes.sizeof(w))
        assert ctx.lib.mp_plan_add(handle, _lib.PLAN_WAIT, ctypes.cast(wb, ctypes.c_void_p), ctypes.sizeof(w), 0) != 0  # waits for itself
        w.waiter_slot, w.signaller_slot = 1, 0
        wb = ctypes.create_string_buffer(bytes(w), ctypes.sizeof(w))
        assert ctx.lib.mp_plan_add(handle, _lib.PLAN_WAIT, ctypes.cast(wb, ctypes.c_void_p), ctypes.sizeof(w), 0) == 0
        assert ctx.lib.mp_plan_size(handle) == 2
        buf.fill_(1.0)
        ctx.check(ctx.lib.mp_plan_run(handle, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "mp_plan_run")
        torch.cuda.synchronize()
        assert float(buf.sum()) == 0.0
    finally:
        ctx.lib.mp_plan_destroy(handle)


def test_hwc_out_without_a_producing_kernel(monkeypatch):
    """hwc_out on a path that has no kernel to write it (stock convolutions): packed from the NCHW
    result instead of raising."""
    from monoport_amd import ops
    from monoport_amd.modeling import backbones
    net = _netg()
    img = torch.from_numpy(syn.synthetic_image(73))[None].to(DEV)
    monkeypatch.setattr(backbones, "ENCODER_CONV", "miopen")
    hwc = torch.empty((1, 128, 128, 256), device=DEV)
    with torch.no_grad():
        outs = net.image_filter(img, last_only=True, hwc_out=hwc)
    assert torch.equal(hwc[0], ops.pack_features(outs[-1][0]))


def test_resnet_filter_dataflow(monkeypatch):
    from monoport_amd.modeling import backbones
    net = backbones.ResnetFilter().eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(shapes, 5).items()})
    net.to(DEV)
    img = torch.from_numpy(syn.synthetic_image(6))[None].to(DEV)
    with torch.no_grad():
        assert net._dataflow_ok(img)
        got = net(img)[0][0]
        monkeypatch.setattr(backbones, "ENCODER_CONV", "miopen")
        ref = net(img)[0][0]
    err = (got - ref).abs().max().item()
    print("ResnetFilter dataflow vs stock ops: %.3g (max|ref| %.3g)" % (err, ref.abs().max().item()))
    assert got.shape == (1, 256, 128, 128) and err <= 1e-4 * max(1.0, ref.abs().max().item())


def test_netc_filter_dataflow_vs_reference_full_coverage():
    """netC.filter(image, feat_prior = netG's last stack) with BOTH encoders on the hand-over kernels
    against the reference's CPU output: the strided samples and every element through the 8 x 8 block /
    pixel means of the fixture."""
    from monoport_amd.modeling import PIFuNetC
    gold = load_golden("encoders")
    netg = _netg(71)
    netc = PIFuNetC().eval()
    shapes = {k: tuple(v.shape) for k, v in netc.image_filter.state_dict().items()}
    netc.image_filter.load_state_dict(
        {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(shapes, 72).items()})
    netc.image_filter.to(DEV)
    img = torch.from_numpy(syn.synthetic_image(73))[None].to(DEV)
    with torch.no_grad():
        fg = netg.filter(img)
        fc = netc.filter(img, feat_prior=fg[-1][-1])
    c0 = fc[0][0][0].cpu().numpy()
    err = float(np.abs(c0[::8, ::8, ::8] - gold["C0"]).max())
    eb, ep = check_full_coverage(gold, "C0", c0)
    print("netC.filter dataflow vs reference: samples %.3g, block means %.3g, pixel means %.3g" % (err, eb, ep))
    assert err <= 1e-4


def test_fused_paths_stay_out_of_autograd():
    """eval mode with gradients requested keeps the differentiable PyTorch ops."""
    from monoport_amd.modeling import backbones
    blk = backbones.ConvBlock(128, 128).to(DEV).eval()
    x = torch.randn((1, 128, 32, 32), device=DEV, requires_grad=True)
    assert not blk._fused_ok(x)
    y = blk(x)
    assert y.grad_fn is not None
    with torch.no_grad():
        assert blk._fused_ok(x.detach())
