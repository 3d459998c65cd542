"""``implicit_seg.functional``: the coarse-to-fine ("lossless octree") reconstruction engine.

``Seg3dLossless`` keeps the constructor and call surface the reference uses
(RTL/main.py:185-195, :392-394) but runs every level on the GPU without touching the host:
upsample + boundary ballot, dilation + compaction, fused PIFu query with scatter
(csrc/octree.hip, csrc/query.hip).

PARITY NOTE: the upstream package is not vendored and not version-pinned by the reference, so the
algorithm here is our restatement of its published scheme; it is checked bit for
bit against the CPU restatement in oracle/pifu_oracle.py and against dense evaluation.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import ops
from ...modeling.MonoPortNet import capture_query
from . import utils  # noqa: F401


class Seg3dLossless(nn.Module):
    def __init__(self, query_func, b_min, b_max, resolutions, channels=1, balance_value=0.5,
                 align_corners=False, visualize=False, debug=False, use_cuda_impl=False,
                 faster=False, use_shadow=False, **kwargs):
        super().__init__()
        self.query_func = query_func
        self.b_min = np.asarray(b_min, np.float32).reshape(-1, 3)
        self.b_max = np.asarray(b_max, np.float32).reshape(-1, 3)
        if self.b_min.shape[0] != 1:
            raise NotImplementedError("batch size 1 (the upstream engine asserts the same)")
        res = []
        for r in resolutions:
            r = np.asarray(r).reshape(-1)
            if r.size == 3 and not (r[0] == r[1] == r[2]):
                raise NotImplementedError("cubic resolutions only")
            res.append(int(r[0]))
        for r in res:
            if r % 2 != 1:
                raise AssertionError("resolution %d need to be odd because of align_corner" % r)
        for a, b in zip(res[:-1], res[1:]):
            if b != 2 * a - 1:
                raise NotImplementedError("resolutions must follow r -> 2r-1 (e.g. 17,33,65,129,257)")
        if channels != 1:
            raise NotImplementedError("one occupancy channel")
        if align_corners:
            raise NotImplementedError("align_corners=False lattice only (the reference's setting)")
        self.resolutions = res
        self.channels = channels
        self.balance_value = float(balance_value)
        self.faster = bool(faster)  # dilation boxes 9/7/3 either way; kept for API parity
        self.use_cuda_impl = use_cuda_impl
        self.last_status = None
        # nn.Module.to(device) is called on the engine (RTL/main.py:195): carry a buffer so it
        # has a device like the upstream module does
        self.register_buffer("_device_tag", torch.zeros(1), persistent=False)

    def forward(self, **kwargs):
        """engine(**kwargs) -> [1,1,R,R,R] f32 occupancy volume (z,y,x) or None when the coarsest
        level has nothing above ``balance_value`` (consumed at RTL/recon.py:32-35)."""
        binding = self._bind(kwargs)
        if binding is None:
            # arbitrary query function: level-at-a-time engine, occupancies from the caller
            volume, counts = ops.recon_generic(self.query_func, kwargs, self._device_tag.device,
                                               self.b_min[0], self.b_max[0], self.resolutions,
                                               self.balance_value)
            self.last_status = torch.tensor([int(volume is not None)] + counts, dtype=torch.int32)
            return None if volume is None else volume[None, None]
        volume, status = self._launch(binding)
        st = status.cpu()  # the one host sync of a reconstruction (upstream syncs per level)
        self.last_status = st
        if int(st[0]) == 0:
            return None
        return volume[None, None]

    def _bind(self, kwargs):
        """Probe ``query_func`` once: if it ends in monoport_amd's MonoPortNet.query (as
        RTL/main.py:169-183 does) return what that call binds, else None."""
        try:
            with capture_query() as cap:
                probe = torch.zeros((1, 1, 3), dtype=torch.float32, device=self._device_tag.device)
                self.query_func(points=probe, **kwargs)
        except Exception:  # noqa: BLE001 -- a foreign function may not like the probe
            return None
        return cap.binding

    def _launch(self, b):
        return ops.recon(b.mlp, b.feat_hwc, b.calib, b.z_scale, self.b_min[0], self.b_max[0],
                         self.resolutions, self.balance_value)

    def forward_async(self, **kwargs):
        """Fused path only, no host sync: (volume [R,R,R], status int32[1+levels]) on device."""
        b = self._bind(kwargs)
        if b is None:
            raise NotImplementedError("forward_async needs a query_func ending in MonoPortNet.query")
        return self._launch(b)


class Seg3dTopk(nn.Module):
    """Imported but never constructed by the reference (RTL/main.py:28)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("Seg3dTopk is unused by the reference pipeline")
