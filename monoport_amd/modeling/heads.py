"""SurfaceClassifier: the skip-connected per-point MLP (mirror of
monoport/lib/modeling/heads/SurfaceClassifier.py).

The parameters live in ``filters.{i}`` Conv1d(k=1) modules so checkpoints load unchanged
(weight [out, in, 1]); the arithmetic runs in the fused HIP query kernel, which reads a copy of
the weights re-packed into MFMA fragment order.  The copy is refreshed automatically whenever a
parameter tensor is replaced or modified in place.
"""
import torch
import torch.nn as nn

from .. import ops

_ACT_CODES = {None: 0, "sigmoid": 1, "tanh": 2}


class SurfaceClassifier(nn.Module):
    def __init__(self, filter_channels, num_views=1, no_residual=False, last_op=None):
        super().__init__()
        if num_views != 1:
            raise NotImplementedError("multi-view averaging (SurfaceClassifier.py:60-66) is unused "
                                      "by the PIFu configs")
        if no_residual:
            raise NotImplementedError("only the skip-concat variant (no_residual=False) is used")
        self.filter_channels = list(filter_channels)
        self.num_views = num_views
        self.no_residual = no_residual
        if isinstance(last_op, nn.Sigmoid):
            last_op = "sigmoid"
        elif isinstance(last_op, nn.Tanh):
            last_op = "tanh"
        if last_op not in _ACT_CODES:
            raise ValueError("last_op must be None, 'sigmoid' or 'tanh'")
        self.last_op = last_op
        c0 = self.filter_channels[0]
        self.filters = nn.ModuleList()
        for l in range(len(self.filter_channels) - 1):
            c_in = self.filter_channels[l] + (c0 if l > 0 else 0)  # SurfaceClassifier.py:26-31
            self.filters.append(nn.Conv1d(c_in, self.filter_channels[l + 1], 1))
        self._packed = None
        self._packed_key = None
        # arithmetic of the MLP GEMMs in the fused HIP kernel: "f32" (exact f32 MFMA, default) or
        # "f16x3" (f32 emulated with three f16 MFMAs per product; netG head only)
        self.precision = "f32"

    # ---- packed-weight cache ------------------------------------------------------------------
    def set_precision(self, precision):
        if precision not in ops.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(ops.PRECISIONS))
        self.precision = precision
        self._packed_key = None  # re-pack on next use
        return self

    def _weights_key(self):
        key = [self.precision]
        for f in self.filters:
            for p in (f.weight, f.bias):
                key.append((p.data_ptr(), p._version, str(p.device)))
        return tuple(key)

    def packed(self):
        """PackedMLP on the parameters' device, rebuilt if the weights changed."""
        key = self._weights_key()
        if self._packed is None or key != self._packed_key:
            dev = self.filters[0].weight.device
            ctx = ops.get_context(dev)
            if self._packed is None or self._packed.ctx is not ctx:
                self._packed = ops.PackedMLP(ctx, self.filter_channels, _ACT_CODES[self.last_op])
            for i, f in enumerate(self.filters):
                self._packed.load_layer(i, f.weight.detach(), f.bias.detach())
            self._packed.set_precision(self.precision)
            self._packed_key = key
        return self._packed

    def forward(self, feature):
        """[B, C_in, N] -> [B, C_out, N] on explicit features (SurfaceClassifier.py:39-71): the same
        fused MFMA kernel as MonoPortNet.query with the gather stage reading the given features
        (the reconstruction path itself never materialises this tensor)."""
        mlp = self.packed()
        return torch.cat([ops.mlp_forward(mlp, feature[b:b + 1]) for b in range(feature.shape[0])], 0)


def PIFuNetGMLP(*args, **kwargs):
    return SurfaceClassifier([257, 1024, 512, 256, 128, 1], 1, False, "sigmoid")


def PIFuNetCMLP(*args, **kwargs):
    return SurfaceClassifier([513, 1024, 512, 256, 128, 3], 1, False, "tanh")
