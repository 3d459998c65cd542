"""``implicit_seg.functional``: the coarse-to-fine ("lossless octree") reconstruction engine.

``Seg3dLossless`` keeps the constructor and call surface the reference uses
(RTL/main.py:185-195, :392-394) but runs every level on the GPU: upsample + boundary ballot,
dilation + compaction, fused PIFu query with scatter (csrc/octree.hip, csrc/query.hip).

PARITY NOTE: the upstream package is not vendored and not version-pinned by the reference, so the
algorithm here is our restatement of its published scheme; it is checked bit for
bit against the CPU restatement in oracle/pifu_oracle.py and against dense evaluation.

Constructor flags (upstream names):
  ``faster=True``   dilation boxes 9^3 / 7^3 / 3^3 by level, no conflict re-examination.  With a
                    ``query_func`` that is a plain MonoPortNet.query call (RTL/main.py:169-183) the
                    whole reconstruction is ONE asynchronous C-ABI call (fused path).
  ``faster=False``  3^3 boxes at every level plus the conflict re-examination loop (nodes whose
                    exact value contradicts the interpolated one get their 3x3x3 neighbourhood
                    evaluated, repeated until none is left); level-at-a-time engine, one host sync
                    per round as upstream.
  ``use_cuda_impl`` upstream switches the implementation of its interpolation, not its results:
                    accepted, both values run the same HIP kernels.
  ``debug``         accepted (upstream only prints timings).
  ``final_level``   (monoport_amd extension, faster=True only) the selection rule of the LAST level:
                    "dilate3" (default) -- the boundary nodes dilated by 3^3 like every level >= 3: the
                    lossless schedule, thresholded volume == thresholded dense evaluation on ordinary
                    bodies; "upstream" -- only nodes whose upsampled inside-mask is exactly 0.5
                    (``is_boundary = valid == 0.5``, no dilation), the rule we recall from the upstream
                    package's faster mode: ~4x fewer points at that level, ~0.4 % of the inside voxels
                    differ from dense evaluation; "interpolate" -- nothing is evaluated at the last
                    level (the other reading of upstream's "last step no examine").  Nothing under the
                    reference pins this; a maintainer who has implicit_seg
                    installed picks the mode that reproduces it.
  ``align_corners=True``, ``visualize=True``, ``use_shadow=True``, ``channels != 1``: not built,
                    NotImplementedError at construction.
"""
import threading
import warnings

import numpy as np
import torch
import torch.nn as nn

from ... import ops
from ...modeling.MonoPortNet import record_query
from . import utils  # noqa: F401


class Seg3dLossless(nn.Module):
    def __init__(self, query_func, b_min, b_max, resolutions, channels=1, balance_value=0.5,
                 align_corners=False, visualize=False, debug=False, use_cuda_impl=False,
                 faster=False, use_shadow=False, final_level="dilate3", validate="always", **kwargs):
        super().__init__()
        if kwargs:  # the upstream constructor swallows **kwargs: stay drop-in, but say so
            warnings.warn("Seg3dLossless: ignoring unknown arguments %s" % sorted(kwargs))
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
        if visualize:
            raise NotImplementedError("visualize=True (upstream's interactive plots) is not built")
        if use_shadow:
            raise NotImplementedError("use_shadow=True is not built (the reference leaves it off)")
        self.resolutions = res
        self.channels = channels
        self.balance_value = float(balance_value)
        self.faster = bool(faster)
        ops._final_level(final_level)  # ValueError for an unknown name
        if final_level != "dilate3" and not self.faster:
            raise NotImplementedError("final_level=%r is a variant of the faster=True schedule" % final_level)
        self.final_level = final_level
        self.use_cuda_impl = bool(use_cuda_impl)  # same kernels either way (see module docstring)
        self.debug = bool(debug)
        self._status = None    # CPU tensor, or the device tensor of a call whose refinement may still be running
        self._early = threading.local()  # per host thread: the EarlyFlags buffers of its fused calls
        self.last_path = None  # "fused" | "generic": which engine served the last call
        # "always" (the default of this drop-in class): every call evaluates the coarsest level through
        # query_func for real and compares it with the fused kernel -- a closure whose arithmetic changes
        # between frames (a flag, `1 - pred` from some frame on) is honoured on the frame it changes; costs
        # one 17^3 query and one host sync per frame (~0.3 ms).
        # "first": the first VALIDATE_CALLS calls are validated; once that many in a row agreed with the
        # fused kernel, later calls with the same network head are trusted and skip the validation query --
        # except every REVALIDATE_EVERY-th one.  For callers that vouch for their query_func
        # (stage_pipeline users that never rebind it).
        if validate not in ("always", "first"):
            raise ValueError("validate must be 'always' or 'first', got %r" % (validate,))
        self.validate = validate
        self._agreed = 0          # consecutive validated calls that agreed
        self._since_check = 0     # trusted calls since the last validated one
        self._trusted_key = None  # (id(packed head), precision, z scale) those calls were bound to
        # nn.Module.to(device) is called on the engine (RTL/main.py:195): carry a buffer so it
        # has a device like the upstream module does
        self.register_buffer("_device_tag", torch.zeros(1), persistent=False)

    @property
    def last_status(self):
        """int32 [1 + levels] on the CPU: (coarsest level non-empty, points queried per level) of the last call.
        A fused call returns as soon as its coarsest level is known (mp_recon_batch_early): reading this waits
        for the call's stream to finish the remaining levels."""
        st = self._status
        if st is not None and st.is_cuda:
            st = self._status = st.cpu()
        return st

    @last_status.setter
    def last_status(self, value):
        self._status = value

    def _early_flags(self, dev, n=1):
        cache = self._early.__dict__.setdefault("flags", {})
        e = cache.get(n)
        if e is None:
            e = cache[n] = ops.EarlyFlags(dev, n)
        return e

    def forward(self, **kwargs):
        """engine(**kwargs) -> [1,1,R,R,R] f32 occupancy volume (z,y,x) or None when the coarsest
        level has nothing above ``balance_value`` (consumed at RTL/recon.py:32-35).

        The coarsest level (17^3 = 4913 points for the reference's settings) is ALWAYS evaluated
        through the caller's ``query_func``, for real.  If that call was exactly one
        MonoPortNet.query and ``faster`` is on, the fused engine runs the whole reconstruction and
        its coarsest level is compared with what ``query_func`` returned; only if they are
        identical is the fused volume returned.  Anything else -- another network, extra
        arithmetic around the call (1 - pred, scaled points, ...), several calls -- goes through
        the level-at-a-time engine, which evaluates every level with ``query_func`` itself.

        Limits of the check: only the coarsest lattice is compared, so a ``query_func`` that equals
        MonoPortNet.query there but post-processes finer levels differently (resolution-dependent
        logic) would pass; and after VALIDATE_CALLS agreeing calls in a row the check is skipped
        for later calls bound to the same head, except every REVALIDATE_EVERY-th one
        (``self.validate = "always"`` keeps it on for every call)."""
        dev = self._device_tag.device
        if (self.faster and self.validate != "always" and self._agreed >= self.VALIDATE_CALLS
                and self._since_check + 1 < self.REVALIDATE_EVERY):
            out = self._forward_trusted(kwargs)
            if out is not NotImplemented:
                self._since_check += 1
                return out
        self._since_check = 0
        eng = ops.LevelEngine(dev, self.b_min[0], self.b_max[0], self.resolutions,
                              self.balance_value, self.faster, self.final_level)
        pts0 = eng.select()
        with record_query() as rec:
            occ0 = self.query_func(points=pts0[None], **kwargs)
        binding = rec.binding if rec.calls == 1 else None
        if binding is not None and self.faster:
            eng.scatter(occ0)  # the caller's values on the coarsest lattice, [r0,r0,r0]
            early = self._early_flags(dev)
            volume, status = ops.recon(binding.mlp, binding.feat_hwc, binding.calib, binding.z_scale,
                                       self.b_min[0], self.b_max[0], self.resolutions,
                                       self.balance_value, final_level=self.final_level, early=early,
                                       expect_level0=eng.cur)
            # the one host sync of a reconstruction (upstream syncs at every level) -- and it waits for the
            # coarsest level only: "None or a volume" and "is query_func the fused kernels' function" are both
            # known there, the finer levels go on refining `volume` on this stream after the call has returned
            nonempty, differs = (int(v) for v in early.wait()[0])
            if differs == 0:
                self.last_status, self.last_path = status, "fused"
                key = self._binding_key(binding)
                self._agreed = self._agreed + 1 if key == self._trusted_key else 1
                self._trusted_key = key
                return None if nonempty == 0 else volume[None, None]
            self._agreed, self._trusted_key = 0, None
            warnings.warn("Seg3dLossless: query_func is not a plain MonoPortNet.query call (its "
                          "values differ from the fused kernel's); using the level-at-a-time engine")
            volume, counts = ops.recon_generic(self.query_func, kwargs, dev, self.b_min[0],
                                               self.b_max[0], self.resolutions, self.balance_value,
                                               self.faster, final_level=self.final_level)
        else:
            volume, counts = ops.recon_generic(self.query_func, kwargs, dev, self.b_min[0],
                                               self.b_max[0], self.resolutions, self.balance_value,
                                               self.faster, level0=(eng, occ0), final_level=self.final_level)
        self.last_path = "generic"
        self.last_status = torch.tensor([int(volume is not None)] + counts, dtype=torch.int32)
        return None if volume is None else volume[None, None]

    VALIDATE_CALLS = 3
    REVALIDATE_EVERY = 32  # a trusted query_func is validated again on every 32nd call

    @staticmethod
    def _binding_key(binding):
        return (id(binding.mlp), binding.mlp.precision, float(binding.z_scale))

    def _forward_trusted(self, kwargs):
        """A query_func whose last VALIDATE_CALLS calls were plain MonoPortNet.query calls agreeing
        with the fused kernel: bind this frame's features / calibration through a one-point probe
        (recorded, not launched) and run the fused engine without the 17^3 validation query (one
        host sync instead of two).  NotImplemented = the binding changed: validate again."""
        probe = torch.zeros((1, 1, 3), dtype=torch.float32, device=self._device_tag.device)
        with record_query(capture_only=True) as rec:
            self.query_func(points=probe, **kwargs)
        b = rec.binding
        if b is None or rec.calls != 1 or self._binding_key(b) != self._trusted_key:
            self._agreed, self._trusted_key = 0, None
            return NotImplemented
        early = self._early_flags(self._device_tag.device)
        volume, status = ops.recon(b.mlp, b.feat_hwc, b.calib, b.z_scale, self.b_min[0], self.b_max[0],
                                   self.resolutions, self.balance_value, final_level=self.final_level, early=early)
        nonempty = int(early.wait()[0, 0])  # waits for the coarsest level only (see forward)
        self.last_status, self.last_path = status, "fused"
        return None if nonempty == 0 else volume[None, None]

    def forward_many(self, kwargs_list):
        """``[self(**kw) for kw in kwargs_list]`` for up to ops.MAX_FRAMES frames at once (monoport_amd extension;
        the hook of a coalescing recon stage, stage_pipeline.Coalesced).  When the engine is in its
        trusted state (see ``forward``) every frame is bound through its one-point probe and ALL of
        them go through one ``mp_recon_batch`` -- every octree level of all frames in one fused-query
        launch, one host sync for all statuses -- with results identical to the per-frame calls bit
        for bit.  In any other state (not yet validated, ``validate = "always"``, a re-validation
        due, a binding that changed) the frames are served one by one by ``forward``."""
        n = len(kwargs_list)
        if n >= 2 and self.validate == "always" and not getattr(self, "_warned_many", False):
            self._warned_many = True
            warnings.warn("Seg3dLossless.forward_many on an engine with validate='always' (the class default) serves "
                          "the frames one by one; construct it with validate='first' to let a coalescing stage batch them")
        if (n < 2 or n > ops.MAX_FRAMES or not self.faster or self.validate == "always" or self._agreed < self.VALIDATE_CALLS
                or self._since_check + n >= self.REVALIDATE_EVERY):
            return [self(**kw) for kw in kwargs_list]
        probe = torch.zeros((1, 1, 3), dtype=torch.float32, device=self._device_tag.device)
        bindings = []
        for kw in kwargs_list:
            with record_query(capture_only=True) as rec:
                self.query_func(points=probe, **kw)
            b = rec.binding
            if b is None or rec.calls != 1 or self._binding_key(b) != self._trusted_key:
                return [self(**kw) for kw in kwargs_list]  # forward() re-validates
            bindings.append(b)
        b0 = bindings[0]
        early = self._early_flags(self._device_tag.device, n)
        volumes, status = ops.recon_batch(b0.mlp, [b.feat_hwc for b in bindings], [b.calib for b in bindings],
                                          b0.z_scale, self.b_min[0], self.b_max[0], self.resolutions,
                                          self.balance_value, final_level=self.final_level, early=early)
        flags = early.wait().clone()  # the one host sync of the whole batch: its coarsest level (see forward)
        self._since_check += n
        self.last_status, self.last_path = status[-1], "fused"
        return [None if int(flags[i, 0]) == 0 else volumes[i][None, None] for i in range(n)]

    def forward_async(self, **kwargs):
        """Fused path only, no host sync and no validation of ``query_func`` (the caller vouches
        that it is a plain MonoPortNet.query call): (volume [R,R,R], status int32[1+levels]) on
        the device."""
        if not self.faster:
            raise NotImplementedError("forward_async is the faster=True schedule")
        probe = torch.zeros((1, 1, 3), dtype=torch.float32, device=self._device_tag.device)
        with record_query(capture_only=True) as rec:
            self.query_func(points=probe, **kwargs)
        b = rec.binding
        if b is None:
            raise NotImplementedError("forward_async needs a query_func ending in MonoPortNet.query")
        return ops.recon(b.mlp, b.feat_hwc, b.calib, b.z_scale, self.b_min[0], self.b_max[0],
                         self.resolutions, self.balance_value, final_level=self.final_level)


class Seg3dTopk(nn.Module):
    """Imported but never constructed by the reference (RTL/main.py:28)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("Seg3dTopk is unused by the reference pipeline")
