"""TEST INFRASTRUCTURE (oracle): Winograd F(2x2, 3x3) restated in numpy float64 -- the algebra csrc/conv_wino.hip
implements for nn.Conv2d(Cin, Cout, 3, 1, 1, bias=False) (the reference's conv3x3, monoport/lib/modeling/backbones/
HGFilters.py:15-19; with reflection padding ResBlkFilters.py:28-84).  Only tests/ may import this.

    Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks")

with d a 4 x 4 input patch, g the 3 x 3 kernel, Y the 2 x 2 output tile, (.) the element-wise product summed over the
input channels.  The three matrices are the ones the kernel's input transform (B^T), weight packing (G) and output
transform (A^T) use."""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def transformed_weights(w):
    """U[i][j][co][ci] = (G g G^T)[i][j] of w [Cout, Cin, 3, 3] (what mp_conv3x3_pack_wino stores, rounded to f32)."""
    return np.einsum("ia,ocab,jb->ijoc", G, np.asarray(w, np.float64), G)


def fragment_index(cout, cin):
    """For every float of the packed buffer (mp_conv3x3_pack_wino's order): the (i, j, co, ci) it holds.
    Fragment ((((rb * 4 + j) * chunks + chunk) * 4 + i) * 2 + g), lane (r = lane & 31, hh = lane >> 5), element e:
    U[i][j][32 rb + r][16 chunk + 8 g + 4 hh + e]."""
    n_chunks = cin // 16
    t = np.arange(16 * cout * cin)
    e, lane = t & 3, (t >> 2) & 63
    q = t >> 8
    g = q & 1
    q = q >> 1
    i = q & 3
    q = q >> 2
    chunk = q % n_chunks
    q = q // n_chunks
    j, rb = q & 3, q >> 2
    return i, j, 32 * rb + (lane & 31), 16 * chunk + 8 * g + 4 * (lane >> 5) + e


def conv3x3(x, w, reflect=False):
    """x [N, Cin, H, W] (H, W even), w [Cout, Cin, 3, 3] -> [N, Cout, H, W] through F(2x2, 3x3), float64."""
    x = np.asarray(x, np.float64)
    n, cin, h, wd = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), mode="reflect" if reflect else "constant")
    u = transformed_weights(w)                                    # [4, 4, Cout, Cin]
    y = np.zeros((n, w.shape[0], h, wd), np.float64)
    for ty in range(h // 2):
        for tx in range(wd // 2):
            d = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]     # [N, Cin, 4, 4]
            v = np.einsum("ir,ncrs,js->ijnc", BT, d, BT)          # B^T d B per channel
            m = np.einsum("ijoc,ijnc->ijno", u, v)                # the 16 GEMMs, summed over Cin
            y[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ri,ijno,sj->nors", AT, m, AT)
    return y
