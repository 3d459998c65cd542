"""The GroupNorm hand-over's accumulator format (csrc/gn_tail.h) restated on the host: 112-bit
fixed-point pairs (value = hi 2^-16 + lo 2^-64) added as integers -- order-independent, exact to
2^-64 per term -- and ops.gn_reference_ss, the host restatement of what a consuming kernel derives from
an accumulator.  No GPU: the kernels themselves are tested in tests/test_encoder_dataflow_gpu.py."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
M64 = 1 << 64


def gn_fixed(x):
    """csrc/gn_tail.h gn_fixed: double -> (hi int64, lo uint64 < 2^48)."""
    xs = x * 65536.0
    fl = math.floor(xs)
    return int(fl), int((xs - fl) * 281474976710656.0)


def accumulate(values, order):
    hi = lo = 0
    for i in order:
        h, l = gn_fixed(float(values[i]))
        hi = (hi + h) % M64   # wrapping 64-bit adds, as the atomics do
        lo = (lo + l) % M64
    return hi, lo


def decode(hi, lo):
    if hi >= 1 << 63:
        hi -= M64
    return hi / 65536.0 + lo / 18446744073709551616.0


def test_fixed_point_sum_is_order_independent_and_exact():
    rs = np.random.RandomState(1)
    # partial sums as workgroups produce them: both signs, magnitudes from 1e-6 to 1e9
    vals = rs.randn(5000) * np.exp(rs.uniform(-14, 21, 5000))
    a = accumulate(vals, range(len(vals)))
    b = accumulate(vals, rs.permutation(len(vals)))
    c = accumulate(vals, reversed(range(len(vals))))
    assert a == b == c                                  # associativity: bit-identical whatever the order
    exact = math.fsum(float(v) for v in vals)
    assert abs(decode(*a) - exact) <= len(vals) * 2.0 ** -48 + abs(exact) * 2.0 ** -52
    for x in (0.0, -0.0, 1.5, -1.5, 2.0 ** -40, -2.0 ** -40, 123456789.125, -6.5e10):
        h, l = gn_fixed(x)
        assert 0 <= l < 1 << 48 and decode(h % M64, l) == pytest.approx(x, abs=2.0 ** -60, rel=2.0 ** -52)


def test_gn_reference_ss_from_replicated_accumulators():
    """Accumulators filled on the host the way producers fill them on the device (per-tile partial
    sums of a tensor, spread over the replicas) -> ops.gn_reference_ss == GroupNorm's definition."""
    from monoport_amd import ops
    g = torch.Generator().manual_seed(3)
    n, c, h, w, reps = 2, 64, 16, 16, 16
    x = torch.randn((n, c, h, w), generator=g, dtype=torch.float64) * 3 + 0.7
    gn = torch.nn.GroupNorm(32, c)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
    acc = np.zeros((reps, n, 32, 4), dtype=np.uint64)
    cpg = c // 32
    tiles = x.reshape(n, 32, cpg, 8, h * w // 8)        # 8 "workgroups" per (image, group)
    for i in range(n):
        for grp in range(32):
            for t in range(8):
                part = tiles[i, grp, :, t]
                for k, v in enumerate((float(part.sum()), float((part * part).sum()))):
                    hi, lo = gn_fixed(v)
                    r = t % reps
                    acc[r, i, grp, 2 * k] = (int(acc[r, i, grp, 2 * k]) + hi) % M64
                    acc[r, i, grp, 2 * k + 1] = (int(acc[r, i, grp, 2 * k + 1]) + lo) % M64
    ss = ops.gn_reference_ss(torch.from_numpy(acc.view(np.int64)), gn, cpg * h * w)
    with torch.no_grad():
        want = gn(x.float())
    got = x.float() * ss[..., 0, None, None] + ss[..., 1, None, None]
    assert (got - want).abs().max().item() <= 2e-5
