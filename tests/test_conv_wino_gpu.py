"""csrc/conv_wino.hip: the encoders' 3x3 convolutions as Winograd F(2x2, 3x3) on f32 MFMA (round 6) -- both kernels
(128 output channels per workgroup, one per CU; 64 per workgroup, two per CU) against the fp64 convolution
(backbones/HGFilters.py:15-19 conv3x3, ResBlkFilters.py:28-84 with reflection padding) and against the direct
kernels of csrc/conv3x3.hip on the same launches: raw output, pyramid-block tail, the statistics handed to the next
GroupNorms, determinism; the packed Winograd-domain weights against G g G^T in fp64; the launcher's choice."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

DIRECT, K64, K128 = 0x400, 0x800, 0x1000


def _gn(c, seed):
    g = torch.Generator().manual_seed(seed)
    gn = torch.nn.GroupNorm(32, c)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        gn.bias.copy_(torch.rand(c, generator=g) - 0.5)
    return gn.to(DEV)


def _acc_of(x):
    from monoport_amd import ops
    ident = torch.zeros((x.shape[0], x.shape[1], 2), device=DEV)
    ident[..., 0] = 1.0
    acc = ops.gn_acc_zeros(DEV, x.shape[0])
    ops.gn_apply(x, ident, relu=False, stats=acc)
    return acc


def _run(tune, x, gn_arg, packed, relu, reflect, res, off, ctot):
    from monoport_amd import _lib, ops
    lib = _lib.load()
    n = x.shape[0]
    out = torch.full((n, ctot, x.shape[2], x.shape[3]), 7.0, device=DEV)
    acc_y, acc_o = ops.gn_acc_zeros(DEV, n), ops.gn_acc_zeros(DEV, n)
    lib.mp_conv3x3_tune(tune)
    try:
        y = ops.conv3x3_fused(x, gn_arg, packed, relu=relu, reflect=reflect, stats=acc_y, out=out, res=res, out_off=off,
                              out_stats=acc_o)
    finally:
        lib.mp_conv3x3_tune(0)
    torch.cuda.synchronize()
    return y, out, acc_y, acc_o


# (N, Cin, Cout, H, W, Ctot, off, reflect)
CASES = [(2, 256, 128, 128, 128, 256, 0, False), (1, 128, 64, 128, 128, 256, 128, False), (3, 64, 64, 64, 64, 256, 192, False),
         (20, 256, 128, 32, 32, 256, 0, False), (4, 128, 128, 64, 64, 256, 128, False), (2, 256, 256, 64, 64, 256, 0, True),
         (2, 16, 64, 32, 32, 128, 64, False), (1, 64, 64, 8, 32, 128, 0, False), (5, 128, 64, 16, 32, 256, 64, True)]


@pytest.mark.parametrize("n,cin,cout,h,w,ctot,off,reflect", CASES)
def test_winograd_kernels_vs_fp64_and_the_direct_kernel(n, cin, cout, h, w, ctot, off, reflect):
    from monoport_amd import ops
    g = torch.Generator().manual_seed(cin * 3 + cout + h + n)
    x = (torch.randn((n, cin, h, w), generator=g) * 2 + 0.3).to(DEV)
    res = torch.randn((n, ctot, h, w), generator=g).to(DEV)
    wt = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(DEV)
    gn_in = _gn(cin, 1) if cin % 32 == 0 else None
    packed = ops.PackedConv3x3(wt)
    assert packed.wino is not None and packed.wino.numel() == 16 * cout * cin
    gn_arg = (_acc_of(x), gn_in) if gn_in is not None else None
    relu = gn_in is not None
    with torch.no_grad():
        v = x.double()
        if gn_in is not None:
            ss = ops.gn_reference_ss(gn_arg[0], gn_in, (cin // 32) * h * w).double()
            v = torch.relu(v * ss[..., 0, None, None] + ss[..., 1, None, None])
        v = torch.nn.ReflectionPad2d(1)(v) if reflect else torch.nn.functional.pad(v, (1, 1, 1, 1))
        ref = torch.nn.functional.conv2d(v, wt.double())
    scale = max(1.0, ref.abs().max().item())
    yd, outd, accyd, accod = _run(DIRECT, x, gn_arg, packed, relu, reflect, res, off, ctot)
    e_direct = (yd.double() - ref).abs().max().item()
    variants = [K64] + ([K128] if cout % 128 == 0 else [])
    for tune in variants:
        y, out, acc_y, acc_o = _run(tune, x, gn_arg, packed, relu, reflect, res, off, ctot)
        err = (y.double() - ref).abs().max().item()
        print("winograd %s %s: max|d| vs fp64 %.3g (direct kernel %.3g, scale %.2g)" % (hex(tune), (n, cin, cout, h, w), err, e_direct, scale))
        assert err <= 1e-5 * scale  # the direct kernel's bar in test_encoder_dataflow_gpu.py is 3e-5
        assert not torch.equal(y, yd)  # it IS the other algorithm
        # pyramid-block tail: exactly y + res on this launch's channels, nothing else written
        assert torch.equal(out[:, off:off + cout], y + res[:, off:off + cout])
        untouched = torch.ones(ctot, dtype=torch.bool)
        untouched[off:off + cout] = False
        assert (out[:, untouched] == 7.0).all()
        # the statistics the next GroupNorms get: the same (scale, shift) as from the direct kernel's
        gy, go = _gn(cout, 2), _gn(ctot, 3)
        ssy = ops.gn_reference_ss(acc_y, gy, (cout // 32) * h * w) - ops.gn_reference_ss(accyd, gy, (cout // 32) * h * w)
        assert ssy.abs().max().item() <= 2e-5
        cpg = ctot // 32
        sso = (ops.gn_reference_ss(acc_o, go, cpg * h * w) - ops.gn_reference_ss(accod, go, cpg * h * w))[:, off:off + cout]
        assert sso.abs().max().item() <= 2e-5
        assert (acc_o[:, :, :off // cpg] == 0).all() and (acc_o[:, :, (off + cout) // cpg:] == 0).all()
        # deterministic (integer statistics, fixed summation order)
        y2, out2, acc_y2, acc_o2 = _run(tune, x, gn_arg, packed, relu, reflect, res, off, ctot)
        assert torch.equal(y, y2) and torch.equal(out, out2) and torch.equal(acc_y, acc_y2) and torch.equal(acc_o, acc_o2)
    if cout % 128 == 0:  # the two Winograd kernels add the same products in the same order
        y64 = _run(K64, x, gn_arg, packed, relu, reflect, res, off, ctot)[0]
        y128 = _run(K128, x, gn_arg, packed, relu, reflect, res, off, ctot)[0]
        assert (y64 - y128).abs().max().item() <= 2e-6 * scale


def test_winograd_weights_are_g_g_gt_rounded_once():
    """mp_conv3x3_pack_wino: U = G g G^T in fp64, rounded to f32, in the fragment order the kernels stream."""
    from monoport_amd import ops
    cout, cin = 64, 32
    g = torch.Generator().manual_seed(3)
    wt = torch.randn((cout, cin, 3, 3), generator=g).to(DEV)
    packed = ops.PackedConv3x3(wt)
    torch.cuda.synchronize()
    up = packed.wino.cpu().numpy()
    from oracle import winograd  # the algebra in float64 (checked against the direct convolution in tests/test_winograd_cpu.py)
    U = winograd.transformed_weights(wt.cpu().numpy()).astype(np.float32)  # [i][j][co][ci], rounded once
    i, j, co, ci = winograd.fragment_index(cout, cin)
    assert np.array_equal(up, U[i, j, co, ci])


def test_winograd_is_chosen_only_where_it_is_built_and_large_enough():
    from monoport_amd import _lib, ops
    lib = _lib.load()
    assert lib.mp_conv3x3_wino_supported(256, 128, 128, 128) == 1 and lib.mp_conv3x3_wino_supported(64, 64, 32, 32) == 1
    assert lib.mp_conv3x3_wino_supported(64, 32, 256, 256) == 0  # 32 output channels: the direct kernels
    assert lib.mp_conv3x3_wino_supported(24, 64, 32, 32) == 0 and lib.mp_conv3x3_wino_supported(64, 64, 12, 32) == 0
    g = torch.Generator().manual_seed(9)
    wt = (torch.randn((64, 64, 3, 3), generator=g) * 0.05).to(DEV)
    packed = ops.PackedConv3x3(wt)
    # a launch of 4 workgroups stays on the direct (split-K) kernel: same bits with and without the Winograd weights
    x = torch.randn((1, 64, 16, 32), generator=g).to(DEV)
    y = ops.conv3x3_fused(x, None, packed, relu=False)
    lib.mp_conv3x3_tune(DIRECT)
    try:
        yd = ops.conv3x3_fused(x, None, packed, relu=False)
    finally:
        lib.mp_conv3x3_tune(0)
    assert torch.equal(y, yd)
    # a misaligned packed_wino pointer is refused (the fragments are read with 128-bit loads)
    a = _lib.Conv3x3Args()
    yb = torch.empty((1, 64, 16, 32), device=DEV)
    a.x, a.n, a.cin, a.h, a.w, a.cout = x.data_ptr(), 1, 64, 16, 32, 64
    a.packed, a.y = packed.data.data_ptr(), yb.data_ptr()
    a.packed_wino = packed.wino.data_ptr() + 4
    ctx = ops.get_encoder_context(torch.device(DEV))
    rc = lib.mp_conv3x3_ex(ctx.handle, ctypes.byref(a), None)
    assert rc != 0 and b"packed_wino" in lib.mp_last_error(ctx.handle)


@pytest.mark.parametrize("batch", [1, 4])
def test_whole_encoders_winograd_vs_direct_kernels(batch):
    """netG.filter (HGFilter, 4 stacks: backbones/HGFilters.py:167-204) with its 3x3 convolutions on the Winograd kernels
    against the same pass on the direct kernels (mp_conv3x3_tune(0x400)): every stack's feature map within 2e-5 -- a
    fifth of the 1e-4 bar the encoders are held to against the reference (tests/test_encoder_dataflow_gpu.py).  At
    batch 1 only the 128^2 launches are large enough for the Winograd kernels; at batch 4 the 64^2 ones as well."""
    import bench
    from monoport_amd import _lib, synthetic as syn
    lib = _lib.load()
    dev = torch.device(DEV)
    netG = bench.build_netg(dev)[0]
    img = torch.stack([torch.from_numpy(syn.synthetic_image(40 + i)) for i in range(batch)]).to(dev)
    with torch.no_grad():
        wino = [f[0].clone() for f in netG.image_filter(img)]
        lib.mp_conv3x3_tune(DIRECT)
        try:
            direct = [f[0].clone() for f in netG.image_filter(img)]
        finally:
            lib.mp_conv3x3_tune(0)
    assert len(wino) == len(direct) == 4
    errs = [(a - b).abs().max().item() for a, b in zip(wino, direct)]
    print("netG.filter batch %d, Winograd vs direct 3x3 kernels per stack: %s" % (batch, ["%.2e" % e for e in errs]))
    assert 0 < max(errs) <= 2e-5
