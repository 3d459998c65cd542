import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pifu_oracle
    pifu_oracle.build()
    return pifu_oracle


def check_full_coverage(g, key, t, tol=2e-5):
    """Every element of a [C,H,W] feature map against the reference fixture: the mean of each 8 x 8
    pixel block per channel and the channel mean of each pixel (oracle/gen_golden.py gen_encoders) --
    a wrong convolution tile cannot hide between the [::8,::8,::8] samples.  A single element off by
    more than 64 * tol, or a tile off by more than tol on average, fails.  Returns the two errors."""
    import numpy as np
    t = np.asarray(t, np.float64)
    c, hh, ww = t.shape
    e_block = float(np.abs(t.reshape(c, hh // 8, 8, ww // 8, 8).mean((2, 4)) - g[key + "_block8"]).max())
    e_pixel = float(np.abs(t.mean(0) - g[key + "_pixel"]).max())
    assert e_block <= tol and e_pixel <= tol, "%s: block %.3g pixel %.3g" % (key, e_block, e_pixel)
    return e_block, e_pixel
