"""oracle/winograd.py (the F(2x2, 3x3) algebra csrc/conv_wino.hip implements) against the direct convolution in float64,
and the packed-weight index map the GPU test (tests/test_conv_wino_gpu.py) checks mp_conv3x3_pack_wino with."""
import numpy as np
import torch

from oracle import winograd


def test_winograd_f23_equals_the_direct_convolution():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 5, 8, 12))
    w = rng.standard_normal((7, 5, 3, 3))
    for reflect in (False, True):
        xt = torch.from_numpy(x)
        xt = torch.nn.ReflectionPad2d(1)(xt) if reflect else torch.nn.functional.pad(xt, (1, 1, 1, 1))
        ref = torch.nn.functional.conv2d(xt, torch.from_numpy(w)).numpy()
        got = winograd.conv3x3(x, w, reflect=reflect)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


def test_the_transform_matrices_are_lavin_and_grays():
    # B^T and A^T are integer matrices, G has only 0, +-1/2, 1: every transform is additions (and a halving folded into
    # the packed weights) -- what lets the kernel keep exact-f32 products
    assert set(np.unique(winograd.BT)) <= {-1.0, 0.0, 1.0} and set(np.unique(winograd.AT)) <= {-1.0, 0.0, 1.0}
    assert set(np.unique(np.abs(winograd.G))) <= {0.0, 0.5, 1.0}
    # a single tap at the centre: the 2 x 2 outputs are the four centre pixels of the patch
    g = np.zeros((1, 1, 3, 3))
    g[0, 0, 1, 1] = 1.0
    d = np.arange(16.0).reshape(1, 1, 4, 4)
    u = winograd.transformed_weights(g)[:, :, 0, 0]
    v = winograd.BT @ d[0, 0] @ winograd.BT.T
    assert np.allclose(winograd.AT @ (u * v) @ winograd.AT.T, d[0, 0, 1:3, 1:3])


def test_the_packed_weight_order_is_a_bijection():
    for cout, cin in ((64, 16), (128, 48)):
        i, j, co, ci = winograd.fragment_index(cout, cin)
        flat = ((i * 4 + j) * cout + co) * cin + ci
        assert flat.size == 16 * cout * cin and np.array_equal(np.sort(flat), np.arange(flat.size))
