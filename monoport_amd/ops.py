"""Tensor-level entry points over the C-ABI (torch is only the allocator / stream provider).

Every function here enqueues hand-written HIP kernels from libmonoport_hip.so on the caller's
current HIP stream and returns ordinary ``torch.Tensor`` objects that outlive the call -- the
ownership rule of the reference's stage pipeline (RTL/dataloader.py:1048-1054).
"""
import ctypes
import os
import threading

import numpy as np
import torch

from . import _lib
from .synthetic import LAST_OP, MLP_DIMS  # noqa: F401  (re-exported for callers)

DIRECTIONS = {"front": 0, "back": 1, "left": 2, "right": 3}  # RTL/recon.py:39-49

_contexts = {}
_contexts_lock = threading.Lock()


class Context:
    """One mp_ctx (calls on it are serialised inside); ``get_context`` keeps two per HIP device, by role."""

    def __init__(self, device_index):
        self.lib = _lib.load()
        self.device_index = int(device_index)
        handle = ctypes.c_void_p()
        rc = self.lib.mp_create(self.device_index, ctypes.byref(handle))
        if rc != 0:
            msg = self.lib.mp_last_error(None)
            raise _lib.MonoportError("mp_create(%d) failed (%d): %s"
                                     % (device_index, rc, msg.decode() if msg else "?"))
        self.handle = handle

    def check(self, rc, what):
        _lib.check(self.handle, rc, what)


def get_context(device, role="query"):
    """Context for a torch device (``cuda:N``).  Raises on CPU tensors: no CPU fallback.

    Two contexts per device (include/monoport_hip.h: "calls on ONE context are serialised by an internal mutex ...
    give each stage its own context"): ``role="query"`` owns the packed heads, the skip-table registry and the
    reconstruction / vertex / render calls; ``role="encoder"`` serves the stateless encoder launches (and the plans
    recorded from them).  The reference runs netG.filter and reconEngine on different host threads
    (RTL/dataloader.py:1026-1053): with one context the filter stage's 137 launches per frame and the recon stage's
    multi-launch mp_recon_batch_early would take turns on one mutex."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.MonoportError(
            "monoport_amd runs on MI355X only (got device %s); there is no CPU path" % device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = idx if role == "query" else (idx, role)
    with _contexts_lock:
        ctx = _contexts.get(key)
        if ctx is None:
            ctx = _contexts[key] = Context(idx)
    return ctx


def get_encoder_context(device):
    return get_context(device, "encoder")


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# mp_mlp_set_precision codes (include/monoport_hip.h)
PRECISIONS = {"f32": 0, "f16x3": 1, "f16w": 2, "f16": 3}


class PackedMLP:
    """Device-resident SurfaceClassifier weights in MFMA fragment order."""

    def __init__(self, ctx, channels, last_op):
        self.ctx = ctx
        self.channels = [int(c) for c in channels]
        self.last_op = int(last_op)
        n = len(self.channels) - 1
        arr = (ctypes.c_int * (n + 1))(*self.channels)
        mid = ctypes.c_int(-1)
        ctx.check(ctx.lib.mp_mlp_create(ctx.handle, n, arr, self.last_op, ctypes.byref(mid)),
                  "mp_mlp_create")
        self.id = mid.value
        self.c = self.channels[0] - 1
        self.cout = self.channels[-1]
        self.precision = "f32"
        self.generation = 0  # bumped by every load_layer: skip tables made before it are stale

    def load_layer(self, layer, weight, bias):
        """weight [out,in] or [out,in,1] (Conv1d k=1), bias [out]; device tensors.  The C side
        forgets the skip tables made with this head (they hold products of the old weights)."""
        w = _f32c(weight.reshape(weight.shape[0], -1))
        b = _f32c(bias)
        self.generation += 1
        self.ctx.check(self.ctx.lib.mp_mlp_load(self.ctx.handle, self.id, layer, _ptr(w), _ptr(b),
                                                w.shape[0], w.shape[1], _stream(w)), "mp_mlp_load")
        # the pack kernels read w/b asynchronously: keep them alive until the stream drains
        w.record_stream(torch.cuda.current_stream(w.device))
        b.record_stream(torch.cuda.current_stream(b.device))

    def set_precision(self, precision):
        """"f32" (default: exact f32 MFMA), "f16x3" (f32 emulated with three f16 MFMAs per
        product), "f16w" (fp16 weights, split activations) or "f16" (fp16 operands); the f16
        variants are netG heads (C = 256) only.  Call after the layers are loaded."""
        code = PRECISIONS[precision]
        self.ctx.check(self.ctx.lib.mp_mlp_set_precision(self.ctx.handle, self.id, code),
                       "mp_mlp_set_precision")
        self.precision = precision

    @classmethod
    def from_layers(cls, device, layers, last_op):
        """layers = [(W[out,in], b[out])] as numpy arrays or tensors (tests / bench fixtures)."""
        ctx = get_context(device)
        ws = [torch.as_tensor(np.asarray(w) if not torch.is_tensor(w) else w) for w, _ in layers]
        dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        mlp = cls(ctx, dims, last_op)
        for i, (w, b) in enumerate(layers):
            mlp.load_layer(i, torch.as_tensor(w).to(device), torch.as_tensor(b).to(device))
        return mlp

    def __del__(self):
        try:
            self.ctx.lib.mp_mlp_destroy(self.ctx.handle, self.id)
        except Exception:
            pass


def pack_features(feats, out=None):
    """[1,Ci,H,W] NCHW maps -> one channels-last [H,W,sum Ci] map (concat order = list order,
    i.e. MonoPortNet.py:44's cat([feat_prior, feat]) when given [prior, feat])."""
    if torch.is_tensor(feats):
        feats = [feats]
    f0 = feats[0]
    ctx = get_context(f0.device)
    h, w = f0.shape[-2], f0.shape[-1]
    total = 0
    for f in feats:
        if f.dim() != 4 or f.shape[0] != 1 or f.shape[-2:] != (h, w):
            raise ValueError("pack_features wants [1,C,H,W] maps of one size, got %s"
                             % (tuple(f.shape),))
        total += f.shape[1]
    if out is None:
        out = torch.empty((h, w, total), dtype=torch.float32, device=f0.device)
    off = 0
    for f in feats:
        fc = _f32c(f)
        ctx.check(ctx.lib.mp_feat_pack_hwc(ctx.handle, _ptr(fc), fc.shape[1], h, w, _ptr(out),
                                           total, off, _stream(out)), "mp_feat_pack_hwc")
        off += fc.shape[1]
    return out


def _calib_dev(calib, device):
    """[B,>=3,4] or [>=3,4] calibration -> contiguous f32 [rows,4] on device (rows 0-2 are read)."""
    c = calib[0] if calib.dim() == 3 else calib
    if c.shape[-1] != 4 or c.shape[0] < 3:
        raise ValueError("calibration must be [>=3,4], got %s" % (tuple(c.shape),))
    return _f32c(c.to(device))


def index(feat_hwc, uv):
    """geometry.py:4-16 on a channels-last map: uv [1,2,N] or [2,N] -> [1,C,N]."""
    ctx = get_context(feat_hwc.device)
    h, w, c = feat_hwc.shape
    u = _f32c(uv.reshape(2, -1))
    n = u.shape[1]
    out = torch.empty((1, c, n), dtype=torch.float32, device=feat_hwc.device)
    ctx.check(ctx.lib.mp_index(ctx.handle, _ptr(feat_hwc), c, h, w, _ptr(u), n, _ptr(out),
                               _stream(out)), "mp_index")
    return out


def orthogonal(points, calib):
    """geometry.py:19-34 (transforms=None): points [1,3,N] -> [1,3,N]."""
    ctx = get_context(points.device)
    p = _f32c(points.reshape(3, -1))
    n = p.shape[1]
    cal = _calib_dev(calib, points.device)
    out = torch.empty((1, 3, n), dtype=torch.float32, device=points.device)
    ctx.check(ctx.lib.mp_orthogonal(ctx.handle, _ptr(p), n, _ptr(cal), _ptr(out), _stream(out)),
              "mp_orthogonal")
    return out


# Skip tables (mp_skip_table): the products of the MLP's weights with the sampled feature (layer 0
# and the skip connections, 42 % of a point's FLOPs) are taken once per texel of a frame's feature
# map instead of once per query point -- 16 GFLOP + 126 MB per frame; the field differs from the
# plain path by f32 rounding only (1-3e-7).  This flag is the default of the callers that make tables
# on their own (pipeline.FrameSlot, MonoPortNet.bind); MONOPORT_SKIP_TABLE=off keeps the plain path.
SKIP_TABLE = os.environ.get("MONOPORT_SKIP_TABLE", "on") != "off"
# MonoPortNet.bind makes a table for a map once it has served this many query points (or at once for
# the octree engine): the table costs what ~16 k points cost on the plain kernels
SKIP_TABLE_MIN_POINTS = int(os.environ.get("MONOPORT_SKIP_TABLE_MIN_POINTS", "16384"))
def table_precision(precision):
    """Whether the fused query of a netG head with this MLP precision blends table rows (mirrors
    launch_query16 in csrc/query16.hip: f16x3 by default, every f16 variant with MONOPORT_TAB16=all, none
    with =off; the exact-f32 kernels always do).  Callers that build tables on their own check it first: a
    table nobody reads costs 16 GFLOP and 128 MB per frame."""
    if precision == "f32":
        return True
    t16 = os.environ.get("MONOPORT_TAB16", "")
    if t16.startswith("a"):
        return True
    if t16.startswith("o"):
        return False
    return precision == "f16x3"


SKIP_TABLE_ROWS = 1952  # kTableRows: the feature segments of layers 0-3 (1024 + 512 + 256 + 128) + the last layer's, padded to 61 cache lines


class SkipTable:
    """A registered skip table: holds the feature map(s) and the table alive and unregisters them
    when released or garbage-collected -- a freed map's address may be handed to the next map, and a
    stale registration would then route that map's queries through this table.  A handle only
    unregisters what is still ITS registration: making a new table for the same map (same buffers,
    next frame) supersedes the old handle."""

    _latest = {}  # (context handle, feature-map address) -> id of the handle that registered it last
    # re-entrant: __del__ -> release() may run from the cyclic GC while this thread holds the lock
    _lock = threading.RLock()

    def __init__(self, ctx, feats, table):
        self.ctx, self.feats, self.table = ctx, list(feats), table
        self._tables = [table[i] for i in range(len(self.feats))] if table.dim() == 4 else [table]
        with SkipTable._lock:
            for f in self.feats:
                SkipTable._latest[(ctx.handle.value if hasattr(ctx.handle, "value") else ctx.handle, f.data_ptr())] = id(self)

    def release(self):
        with SkipTable._lock:
            for f, t in zip(self.feats, self._tables):
                key = (self.ctx.handle.value if hasattr(self.ctx.handle, "value") else self.ctx.handle, f.data_ptr())
                if SkipTable._latest.get(key) == id(self):
                    del SkipTable._latest[key]
                    self.ctx.lib.mp_skip_table_release(self.ctx.handle, _ptr(f), _ptr(t))
            self.feats, self._tables = [], []

    def __del__(self):
        try:
            self.release()
        except Exception:  # interpreter shutdown
            pass


def skip_table(mlp, feat_hwc, out=None):
    """mp_skip_table: the skip table of a channels-last feature map [H,W,256] for a netG head --
    table[y,x,:] = the products of every layer's feature-segment weights with feat[y,x,:]
    (SurfaceClassifier's layer 0 and the skip connections of layers 1-4: 1921 rows) -- computed
    once per map and REGISTERED for it: every later fused query (query / recon / recon_batch ...)
    on maps that all have a table blends four table rows per point instead of multiplying those
    weights with the sampled feature on the MFMAs (42 % of a point's FLOPs).  The result differs
    from the plain path by f32 rounding only.  Returns a SkipTable handle (``.table`` is the
    [H,W,1952] tensor); the registration lasts until ``.release()`` or the handle's collection.
    Call again after rewriting the feature map."""
    ctx = mlp.ctx
    h, w, c = feat_hwc.shape
    if out is None:
        out = torch.empty((h, w, SKIP_TABLE_ROWS), dtype=torch.float32, device=feat_hwc.device)
    elif tuple(out.shape) != (h, w, SKIP_TABLE_ROWS) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("skip_table: out must be a contiguous float32 [%d,%d,%d]" % (h, w, SKIP_TABLE_ROWS))
    ctx.check(ctx.lib.mp_skip_table(ctx.handle, mlp.id, _ptr(feat_hwc), c, h, w, _ptr(out), _stream(out)),
              "mp_skip_table")
    return SkipTable(ctx, [feat_hwc], out)


def skip_table_batch(mlp, feat_hwc_all, out=None):
    """mp_skip_table_batch: the tables of B maps stored back to back [B,H,W,256] -> [B,H,W,1952] in
    one launch; each map feat_hwc_all[i] is registered with its table out[i].  Returns a SkipTable
    handle for all of them."""
    ctx = mlp.ctx
    b, h, w, c = feat_hwc_all.shape
    if not feat_hwc_all.is_contiguous():
        raise ValueError("skip_table_batch: the maps must be contiguous")
    if out is None:
        out = torch.empty((b, h, w, SKIP_TABLE_ROWS), dtype=torch.float32, device=feat_hwc_all.device)
    elif tuple(out.shape) != (b, h, w, SKIP_TABLE_ROWS) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("skip_table_batch: out must be a contiguous float32 [%d,%d,%d,%d]"
                         % (b, h, w, SKIP_TABLE_ROWS))
    ctx.check(ctx.lib.mp_skip_table_batch(ctx.handle, mlp.id, b, _ptr(feat_hwc_all), c, h, w, _ptr(out),
                                          _stream(out)), "mp_skip_table_batch")
    return SkipTable(ctx, [feat_hwc_all[i] for i in range(b)], out)


def skip_table_release(ctx, feat_hwc=None):
    """Forget the table registered for feat_hwc (None: every table of the context), whichever handle
    made it."""
    with SkipTable._lock:
        h = ctx.handle.value if hasattr(ctx.handle, "value") else ctx.handle
        for key in [k for k in SkipTable._latest if k[0] == h and (feat_hwc is None or k[1] == feat_hwc.data_ptr())]:
            del SkipTable._latest[key]
    ctx.check(ctx.lib.mp_skip_table_release(ctx.handle, _ptr(feat_hwc) if feat_hwc is not None else None, None),
              "mp_skip_table_release")


def query(mlp, feat_hwc, points, calib, z_scale):
    """MonoPortNet.query (eval, one stage).  points [1,3,N] with ANY strides (the permuted view
    query_func builds at RTL/main.py:176-177 is consumed in place) -> [1,Cout,N]."""
    ctx = mlp.ctx
    if points.dim() != 3 or points.shape[0] != 1 or points.shape[1] != 3:
        raise ValueError("points must be [1,3,N], got %s" % (tuple(points.shape),))
    if points.dtype != torch.float32:
        points = points.float()
    h, w, c = feat_hwc.shape
    n = points.shape[2]
    cal = _calib_dev(calib, feat_hwc.device)
    out = torch.empty((1, mlp.cout, n), dtype=torch.float32, device=feat_hwc.device)
    ctx.check(ctx.lib.mp_query(ctx.handle, mlp.id, _ptr(feat_hwc), c, h, w, _ptr(points), n,
                               points.stride(2), points.stride(1), _ptr(cal), float(z_scale),
                               _ptr(out), _stream(out)), "mp_query")
    return out


def mlp_forward(mlp, feature):
    """SurfaceClassifier.forward: feature [1,C+1,N] -> [1,Cout,N] (mp_mlp_forward)."""
    ctx = mlp.ctx
    if feature.dim() != 3 or feature.shape[0] != 1 or feature.shape[1] != mlp.c + 1:
        raise ValueError("feature must be [1,%d,N], got %s" % (mlp.c + 1, tuple(feature.shape)))
    f = _f32c(feature)
    n = f.shape[2]
    out = torch.empty((1, mlp.cout, n), dtype=torch.float32, device=f.device)
    ctx.check(ctx.lib.mp_mlp_forward(ctx.handle, mlp.id, _ptr(f), n, _ptr(out), _stream(f)),
              "mp_mlp_forward")
    return out


def query_counted(mlp, feat_hwc, points, count, calib, z_scale, out=None):
    """mp_query_counted: points [3,cap] contiguous, count int32[1] on device -> [Cout,cap]."""
    ctx = mlp.ctx
    h, w, c = feat_hwc.shape
    cap = points.shape[1]
    cal = _calib_dev(calib, feat_hwc.device)
    if out is None:
        out = torch.zeros((mlp.cout, cap), dtype=torch.float32, device=feat_hwc.device)
    ctx.check(ctx.lib.mp_query_counted(ctx.handle, mlp.id, _ptr(feat_hwc), c, h, w, _ptr(points),
                                       cap, _ptr(count), _ptr(cal), float(z_scale), _ptr(out),
                                       _stream(out)), "mp_query_counted")
    return out


def query_counted_batch(mlp, feats_hwc, points, counts, calibs, z_scale, outs=None):
    """mp_query_counted_batch: one fused-query launch for up to MAX_FRAMES frames.  feats_hwc / points
    ([3,cap] each, one cap) / counts (int32[1] each) / calibs: lists of per-frame device tensors
    -> list of [Cout,cap]."""
    ctx = mlp.ctx
    n = len(feats_hwc)
    h, w, c = feats_hwc[0].shape
    cap = points[0].shape[1]
    dev = feats_hwc[0].device
    cals = [_calib_dev(cb, dev) for cb in calibs]
    if outs is None:
        outs = [torch.zeros((mlp.cout, cap), dtype=torch.float32, device=dev) for _ in range(n)]
    for f, p in zip(feats_hwc, points):
        assert f.shape == (h, w, c) and f.is_contiguous() and p.shape == (3, cap) and p.is_contiguous()
    ptrs = ctypes.c_void_p * n
    ctx.check(ctx.lib.mp_query_counted_batch(
        ctx.handle, mlp.id, n, ptrs(*[f.data_ptr() for f in feats_hwc]), c, h, w,
        ptrs(*[p.data_ptr() for p in points]), cap, ptrs(*[k.data_ptr() for k in counts]),
        ptrs(*[cb.data_ptr() for cb in cals]), float(z_scale), ptrs(*[o.data_ptr() for o in outs]),
        _stream(outs[0])), "mp_query_counted_batch")
    stream = torch.cuda.current_stream(dev)
    for t in cals:
        t.record_stream(stream)
    return outs


# Selection rule of the LAST octree level (include/monoport_hip.h, MP_FINAL_*; Seg3dLossless docstring)
FINAL_LEVELS = {"dilate3": 0, "upstream": 1, "interpolate": 2}
MAX_FRAMES = _lib.load().mp_max_frames()  # kMaxFrames: frames per mp_recon_batch / mp_query_counted_batch call


class EarlyFlags:
    """Buffers of one mp_recon_batch_early hand-over for up to ``n`` frames: device flags, their pinned host copy
    and the event recorded behind the copy.  ``wait()`` blocks until the coarsest octree level of the call is
    done (~0.1 ms of GPU time into it) and returns [n,2] int32: (level 0 non-empty, level-0 values differ from the
    expected ones).  One object serves one call at a time (reuse it after ``wait``)."""

    def __init__(self, device, n=1):
        self.n = int(n)
        self.dev = torch.zeros(2 * self.n, dtype=torch.int32, device=device)
        self.host = torch.zeros(2 * self.n, dtype=torch.int32).pin_memory()
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(device))  # materialises the hipEvent_t the C side re-records

    def struct(self, expect):
        n = self.n
        self._expect = None
        if expect is not None:
            self._expect = (ctypes.c_void_p * n)(*[None if e is None else e.data_ptr() for e in expect])
        return _lib.ReconEarly(ctypes.cast(self._expect, ctypes.c_void_p) if self._expect is not None else None,
                               self.dev.data_ptr(), self.host.data_ptr(), self.event.cuda_event)

    def wait(self):
        self.event.synchronize()
        return self.host.view(self.n, 2)


def _final_level(final_level):
    try:
        return FINAL_LEVELS[final_level]
    except KeyError:
        raise ValueError("final_level must be one of %s, got %r" % (sorted(FINAL_LEVELS), final_level)) from None


def recon(mlp, feat_hwc, calib, z_scale, b_min, b_max, resolutions, balance=0.5, volume=None,
          status=None, final_level="dilate3", early=None, expect_level0=None):
    """Coarse-to-fine occupancy volume (Seg3dLossless replacement).  Returns (volume [R,R,R]
    f32, status int32[1+levels]) -- both on device, nothing synchronised.  ``early`` / ``expect_level0``: see
    ``recon_batch``."""
    st = None if status is None else status.reshape(1, -1)
    volumes, st = recon_batch(mlp, [feat_hwc], [calib], z_scale, b_min, b_max, resolutions, balance,
                              None if volume is None else [volume], st, final_level, early,
                              None if expect_level0 is None else [expect_level0])
    return volumes[0], st[0]


def recon_batch(mlp, feats_hwc, calibs, z_scale, b_min, b_max, resolutions, balance=0.5,
                volumes=None, status=None, final_level="dilate3", early=None, expect_level0=None):
    """``recon`` over up to MAX_FRAMES independent frames in one call: every octree level evaluates the
    selected nodes of all frames in ONE fused-query launch (the coarse levels of a single frame
    cannot fill 256 CUs).  feats_hwc: list of [H,W,C] maps; calibs: [B,4,4] (or list of [1,4,4]);
    volumes: list of [R,R,R]; status: [B, 1+levels] int32.  Results equal B separate ``recon``
    calls bit for bit.  ``early``: an ``EarlyFlags`` for B frames -- mp_recon_batch_early: after the coarsest level
    the call hands (non-empty, differs-from-``expect_level0[b]``) per frame to the host (``early.wait()``) and goes
    on refining; ``expect_level0``: list of [r0,r0,r0] f32 tensors (or None entries)."""
    ctx = mlp.ctx
    n = len(feats_hwc)
    h, w, c = feats_hwc[0].shape
    res = [int(r) for r in resolutions]
    r_last = res[-1]
    dev = feats_hwc[0].device
    if torch.is_tensor(calibs):
        calibs = [calibs[b:b + 1] for b in range(n)]
    cals = [_calib_dev(cb, dev) for cb in calibs]
    if volumes is None:
        volumes = [torch.empty((r_last, r_last, r_last), dtype=torch.float32, device=dev)
                   for _ in range(n)]
    if status is None:
        status = torch.empty((n, 1 + len(res)), dtype=torch.int32, device=dev)
    assert status.is_contiguous() and status.shape == (n, 1 + len(res))
    for f in feats_hwc:
        assert f.shape == (h, w, c) and f.is_contiguous() and f.dtype == torch.float32
    bmin = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(b_min, np.float32).reshape(3)])
    bmax = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(b_max, np.float32).reshape(3)])
    res_c = (ctypes.c_int * len(res))(*res)
    ptrs = ctypes.c_void_p * n
    if early is not None:
        if early.n != n:
            raise ValueError("recon_batch: EarlyFlags for %d frames, call has %d" % (early.n, n))
        if expect_level0 is not None:
            for e in expect_level0:
                assert e is None or (e.numel() == res[0] ** 3 and e.is_contiguous() and e.dtype == torch.float32)
        est = early.struct(expect_level0)
        early_arg = ctypes.byref(est)
    else:
        early_arg = None
    ctx.check(ctx.lib.mp_recon_batch_early(
        ctx.handle, mlp.id, n, ptrs(*[f.data_ptr() for f in feats_hwc]), c, h, w,
        ptrs(*[cb.data_ptr() for cb in cals]), float(z_scale), bmin, bmax, res_c, len(res),
        float(balance), _final_level(final_level), ptrs(*[v.data_ptr() for v in volumes]),
        ptrs(*[status[b].data_ptr() for b in range(n)]), early_arg, _stream(volumes[0])), "mp_recon_batch_early")
    stream = torch.cuda.current_stream(dev)
    if expect_level0 is not None:
        for e in expect_level0:
            if e is not None:
                e.record_stream(stream)
    for t in cals:
        t.record_stream(stream)
    return volumes, status


class LevelEngine:
    """The coarse-to-fine engine one step at a time, for an ARBITRARY ``query_func``: node
    selection, lattice coordinates, conflict detection and scatter run as HIP kernels
    (csrc/octree.hip), the occupancies come from the caller; one host sync per step for the point
    count (as the upstream engine).  ``faster=True``: dilation boxes 9/7/3 by level, no conflict
    re-examination; ``faster=False``: 3^3 boxes at every level and, after each evaluation, the
    3x3x3 neighbourhoods of nodes whose exact value contradicts the interpolated one are evaluated
    too, until no contradiction is left."""

    def __init__(self, device, b_min, b_max, resolutions, balance=0.5, faster=True, final_level="dilate3"):
        self.final_level = final_level
        _final_level(final_level)
        self.ctx = get_context(device)
        self.dev = torch.device(device)
        self.res = [int(r) for r in resolutions]
        self.rf = self.res[-1]
        self.balance = float(balance)
        self.faster = bool(faster)
        self.bmin = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(b_min, np.float32).reshape(3)])
        self.bmax = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(b_max, np.float32).reshape(3)])
        self.count = torch.zeros((1,), dtype=torch.int32, device=self.dev)
        self.level = -1
        self.prev = self.ev_prev = None
        self.cur = self.ev_cur = self.packed = None
        self.counts = []       # points queried per level (conflict rounds included)
        self.rounds = []       # conflict rounds per level

    def _points(self, packed, n, r):
        pts = torch.empty((n, 3), dtype=torch.float32, device=self.dev)
        ctx = self.ctx
        ctx.check(ctx.lib.mp_lattice_points(ctx.handle, _ptr(packed), _ptr(self.count), n,
                                            (self.rf - 1) // (r - 1), self.rf, self.bmin, self.bmax,
                                            _ptr(pts), _stream(pts)), "mp_lattice_points")
        return pts

    def select(self):
        """Advance to the next level: returns the [n,3] world points to evaluate (n may be 0)."""
        self.level += 1
        level, r = self.level, self.res[self.level]
        ctx, dev = self.ctx, self.dev
        words = r * r * ((r + 63) // 64)
        if level > 0:
            self.prev, self.ev_prev = self.cur, self.ev_cur
        self.cur = torch.empty((r, r, r), dtype=torch.float32, device=dev)
        self.ev_cur = torch.empty((words,), dtype=torch.int64, device=dev)
        bnd = torch.empty((words,), dtype=torch.int64, device=dev)
        self.packed = torch.empty((r ** 3,), dtype=torch.int32, device=dev)
        box = 3 if not self.faster else {1: 9, 2: 7}.get(level, 3)
        if self.faster and level == len(self.res) - 1 and level > 0:
            # the last level's rule (mp_octree_select_box: 1 = upsampled mask == 0.5, undilated; 0 = none)
            box = {"dilate3": box, "upstream": 1, "interpolate": 0}[self.final_level]
        ctx.check(ctx.lib.mp_octree_select_box(
            ctx.handle, _ptr(self.prev) if level > 0 else None, self.res[level - 1] if level > 0 else 0,
            _ptr(self.cur), r, _ptr(self.ev_prev) if level > 0 else None, _ptr(self.ev_cur),
            _ptr(bnd), box, self.balance, _ptr(self.packed), _ptr(self.count), _stream(self.cur)),
            "mp_octree_select_box")
        self.n = int(self.count.item())
        self.counts.append(self.n)
        self.rounds.append(0)
        return self._points(self.packed, self.n, r) if self.n else None

    def _values(self, occ, n):
        if isinstance(occ, (list, tuple)):
            occ = torch.stack(list(occ))
        vals = _f32c(occ.reshape(-1))
        if vals.shape[0] != n:
            raise ValueError("query_func returned %d values for %d points" % (vals.shape[0], n))
        return vals

    def scatter(self, occ):
        """Hand over the occupancies of the points ``select`` (or the previous ``scatter``)
        returned.  Returns the next batch of points of THIS level to evaluate (faster=False:
        neighbourhoods of conflicting nodes) or None when the level is complete."""
        ctx, r, n = self.ctx, self.res[self.level], self.n
        vals = self._values(occ, n)
        nxt = None
        if not self.faster and self.level > 0:
            nxt = torch.empty((r ** 3,), dtype=torch.int32, device=self.dev)
            nxt_count = torch.zeros((1,), dtype=torch.int32, device=self.dev)
            ctx.check(ctx.lib.mp_octree_conflicts(
                ctx.handle, _ptr(self.packed), _ptr(self.count), n, r, _ptr(vals), _ptr(self.cur),
                self.balance, _ptr(self.ev_cur), _ptr(nxt), _ptr(nxt_count), _stream(vals)),
                "mp_octree_conflicts")
        ctx.check(ctx.lib.mp_scatter_nodes(ctx.handle, _ptr(self.packed), _ptr(self.count), n, r,
                                           _ptr(vals), _ptr(self.cur), _stream(self.cur)),
                  "mp_scatter_nodes")
        if nxt is None:
            return None
        m = int(nxt_count.item())
        if m == 0:
            return None
        # claimed through atomics: sort for a reproducible evaluation order
        self.packed = torch.sort(nxt[:m])[0].contiguous()
        self.count = nxt_count
        self.n = m
        self.counts[-1] += m
        self.rounds[-1] += 1
        return self._points(self.packed, m, r)

    def empty(self):
        """Level 0 only: nothing above the threshold (the engine then returns None)."""
        return not bool((self.cur > self.balance).any())


def recon_generic(query_func, kwargs, device, b_min, b_max, resolutions, balance=0.5, faster=True,
                  level0=None, final_level="dilate3"):
    """Seg3dLossless for an ARBITRARY ``query_func(points=[1,N,3], **kwargs) -> [1,1,N]`` on top
    of ``LevelEngine``.  ``level0`` = (engine, occupancies) when the caller has already evaluated
    the coarsest level through the engine.  Returns (volume [R,R,R] or None, per-level counts)."""
    if level0 is None:
        eng = LevelEngine(device, b_min, b_max, resolutions, balance, faster, final_level)
        pts = eng.select()
        occ = query_func(points=pts[None], **kwargs)
    else:
        eng, occ = level0
    eng.scatter(occ)
    if eng.empty():
        return None, eng.counts
    for _ in range(1, len(eng.res)):
        pts = eng.select()
        while pts is not None:
            pts = eng.scatter(query_func(points=pts[None], **kwargs))
    return eng.cur, eng.counts


def stream_release(stream):
    """Free the scratch arena the context keeps for ``stream`` (a torch.cuda.Stream); call it when
    a stream that made C-ABI calls is retired.  Synchronises the device."""
    for ctx in (get_context(stream.device), get_encoder_context(stream.device)):
        ctx.check(ctx.lib.mp_stream_release(ctx.handle, ctypes.c_void_p(stream.cuda_stream)),
                  "mp_stream_release")


# ---- recorded launch sequences (mp_plan_*, csrc/plan.hip) --------------------------------------------
_plan_tls = threading.local()


class Plan:
    """A recorded sequence of encoder launches (mp_plan): ``run()`` replays it on the current stream with ONE
    foreign call.  Keeps every tensor the commands point at alive."""

    def __init__(self, ctx, handle, keep, n_cmds):
        self.ctx, self.handle, self.keep, self.n_cmds = ctx, handle, keep, n_cmds

    def run(self, device):
        st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        self.ctx.check(self.ctx.lib.mp_plan_run(self.handle, st), "mp_plan_run")

    def __del__(self):
        try:
            self.ctx.lib.mp_plan_destroy(self.handle)
        except Exception:  # interpreter shutdown
            pass


class record_plan:
    """Context manager (per host thread): the encoder wrappers below (convk, gn_apply, conv3x3_fused,
    conv1x1_fused, avgpool2_gn, upsample_add_gn, GnArena, plan_wait) run normally AND append what they
    launched to a plan; ``.finish()`` builds it.  The stream that is current on entry is slot 0, other
    streams get slots in order of first use."""

    def __init__(self, device, keep_alive=True):
        """``keep_alive=False``: the caller guarantees the lifetime of every buffer itself (a private
        allocator pool that outlives the plan); the plan then holds no tensor references, so the pass it is
        recorded from recycles its intermediates as a launch-by-launch pass does."""
        self.ctx = get_encoder_context(device)
        self.device = torch.device(device)
        self.keep_alive = bool(keep_alive)
        self.cmds, self.keep = [], []
        self.slots = {torch.cuda.current_stream(self.device).cuda_stream: 0}

    def __enter__(self):
        if getattr(_plan_tls, "rec", None) is not None:
            raise RuntimeError("record_plan does not nest")
        _plan_tls.rec = self
        return self

    def __exit__(self, *exc):
        _plan_tls.rec = None
        return False

    def slot(self, stream=None):
        h = (stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream
        if h not in self.slots:
            self.slots[h] = len(self.slots)
        return self.slots[h]

    def add(self, kind, struct, tensors=()):
        self.cmds.append((kind, bytes(struct), self.slot()))
        if self.keep_alive:
            self.keep.extend(t for t in tensors if t is not None)

    def finish(self):
        ctx = self.ctx
        handle = ctypes.c_void_p()
        ctx.check(ctx.lib.mp_plan_create(ctx.handle, len(self.slots) - 1, ctypes.byref(handle)), "mp_plan_create")
        plan = Plan(ctx, handle, self.keep, len(self.cmds))
        for kind, blob, slot in self.cmds:
            buf = ctypes.create_string_buffer(blob, len(blob))
            ctx.check(ctx.lib.mp_plan_add(handle, kind, ctypes.cast(buf, ctypes.c_void_p), len(blob), slot),
                      "mp_plan_add")
        return plan


def _recording():
    return getattr(_plan_tls, "rec", None)


def plan_wait(waiter, signaller):
    """``waiter.wait_stream(signaller)`` (torch streams) that a plan being recorded remembers."""
    waiter.wait_stream(signaller)
    rec = _recording()
    if rec is not None:
        w = _lib.PlanWaitArgs()
        w.waiter_slot, w.signaller_slot = rec.slot(waiter), rec.slot(signaller)
        rec.cmds.append((_lib.PLAN_WAIT, bytes(w), 0))


def forward_vertices_raw(volume, direction="front"):
    """mp_forward_vertices: returns capacity-sized (X, Y, Z, norm, count) device tensors."""
    vol = volume
    while vol.dim() > 3:
        vol = vol[0]
    vol = _f32c(vol)
    r = vol.shape[2]
    if vol.shape != (r, r, r):
        raise ValueError("forward_vertices wants a cubic volume, got %s" % (tuple(vol.shape),))
    ctx = get_context(vol.device)
    cap = r * r
    dev = vol.device
    x = torch.empty((cap,), dtype=torch.int64, device=dev)
    y = torch.empty((cap,), dtype=torch.int64, device=dev)
    z = torch.empty((cap,), dtype=torch.float32, device=dev)
    nrm = torch.empty((cap, 3), dtype=torch.float32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    ctx.check(ctx.lib.mp_forward_vertices(ctx.handle, _ptr(vol), r, DIRECTIONS[direction], _ptr(x),
                                          _ptr(y), _ptr(z), _ptr(nrm), _ptr(count), _stream(vol)),
              "mp_forward_vertices")
    return x, y, z, nrm, count


def forward_vertices_raw_batch(volumes, direction="front"):
    """mp_forward_vertices_batch: ``[forward_vertices_raw(v, direction) for v in volumes]`` (up to MAX_FRAMES cubic
    volumes of one size) in one set of launches; every frame's (X, Y, Z, norm, count) are views of five tensors."""
    vols = []
    for v in volumes:
        while v.dim() > 3:
            v = v[0]
        vols.append(_f32c(v))
    n = len(vols)
    r = vols[0].shape[2]
    if any(tuple(v.shape) != (r, r, r) for v in vols):
        raise ValueError("forward_vertices_raw_batch wants cubic volumes of one size")
    dev = vols[0].device
    ctx = get_context(dev)
    cap = r * r
    x = torch.empty((n, cap), dtype=torch.int64, device=dev)
    y = torch.empty((n, cap), dtype=torch.int64, device=dev)
    z = torch.empty((n, cap), dtype=torch.float32, device=dev)
    nrm = torch.empty((n, cap, 3), dtype=torch.float32, device=dev)
    count = torch.empty((n, 1), dtype=torch.int32, device=dev)
    out = []
    for f0 in range(0, n, MAX_FRAMES):
        f1 = min(f0 + MAX_FRAMES, n)
        ptrs = ctypes.c_void_p * (f1 - f0)
        ctx.check(ctx.lib.mp_forward_vertices_batch(
            ctx.handle, f1 - f0, ptrs(*[vols[f].data_ptr() for f in range(f0, f1)]), r, DIRECTIONS[direction],
            ptrs(*[x[f].data_ptr() for f in range(f0, f1)]), ptrs(*[y[f].data_ptr() for f in range(f0, f1)]),
            ptrs(*[z[f].data_ptr() for f in range(f0, f1)]), ptrs(*[nrm[f].data_ptr() for f in range(f0, f1)]),
            ptrs(*[count[f].data_ptr() for f in range(f0, f1)]), _stream(x)), "mp_forward_vertices_batch")
    stream = torch.cuda.current_stream(dev)
    for v in vols:
        v.record_stream(stream)
    for f in range(n):
        out.append((x[f], y[f], z[f], nrm[f], count[f]))
    return out


def paint_batch(xs, ys, values, channel_major, counts, res, scale, bias, lo, hi):
    """mp_paint_batch: ``[paint(x, y, v, channel_major, c, res, ...) for ...]`` (up to MAX_FRAMES renders of one size,
    one capacity) in two launches; the images are views of one [n, res, res, 3] tensor."""
    n = len(xs)
    dev = xs[0].device
    ctx = get_context(dev)
    cap = xs[0].shape[0]
    vals = [_f32c(v) for v in values]
    images = torch.empty((n, res, res, 3), dtype=torch.float32, device=dev)
    for f0 in range(0, n, MAX_FRAMES):
        f1 = min(f0 + MAX_FRAMES, n)
        ptrs = ctypes.c_void_p * (f1 - f0)
        ctx.check(ctx.lib.mp_paint_batch(
            ctx.handle, f1 - f0, ptrs(*[xs[f].data_ptr() for f in range(f0, f1)]),
            ptrs(*[ys[f].data_ptr() for f in range(f0, f1)]), ptrs(*[vals[f].data_ptr() for f in range(f0, f1)]),
            int(channel_major), ptrs(*[counts[f].data_ptr() for f in range(f0, f1)]), cap, int(res), float(scale),
            float(bias), float(lo), float(hi), ptrs(*[images[f].data_ptr() for f in range(f0, f1)]), _stream(images)),
            "mp_paint_batch")
    stream = torch.cuda.current_stream(dev)
    for v in vals:
        v.record_stream(stream)
    return [images[f] for f in range(n)]


def vertex_points(x, y, z, count, res, mat):
    """(X, Y, res - Z) through the voxel->world matrix (RTL/main.py:231-237) -> [3,cap]."""
    ctx = get_context(x.device)
    cap = x.shape[0]
    m = (ctypes.c_float * 16)(*[float(v) for v in np.asarray(mat, np.float32).reshape(16)])
    pts = torch.zeros((3, cap), dtype=torch.float32, device=x.device)
    ctx.check(ctx.lib.mp_vertex_points(ctx.handle, _ptr(x), _ptr(y), _ptr(z), _ptr(count), cap,
                                       int(res), m, _ptr(pts), _stream(pts)), "mp_vertex_points")
    return pts


def paint(x, y, values, channel_major, count, res, scale, bias, lo, hi):
    """canvas of ones [res,res,3] with image[X,Y,:] = clamp(values*scale+bias) (main.py:220-248)."""
    ctx = get_context(x.device)
    cap = x.shape[0]
    image = torch.empty((res, res, 3), dtype=torch.float32, device=x.device)
    values = _f32c(values)
    ctx.check(ctx.lib.mp_paint(ctx.handle, _ptr(x), _ptr(y), _ptr(values), int(channel_major),
                               _ptr(count), cap, int(res), float(scale), float(bias), float(lo),
                               float(hi), _ptr(image), _stream(image)), "mp_paint")
    return image


def visualize(image, size=256):
    """mp_visualize: ([size,size,3] f32 = 255 * rot90 + nearest resize of ``image`` [res,res,3],
    [size,size] uint8 foreground mask) on device."""
    ctx = get_context(image.device)
    img = _f32c(image)
    res = img.shape[0]
    out = torch.empty((size, size, 3), dtype=torch.float32, device=img.device)
    mask = torch.empty((size, size), dtype=torch.uint8, device=img.device)
    ctx.check(ctx.lib.mp_visualize(ctx.handle, _ptr(img), res, int(size), _ptr(out), _ptr(mask),
                                   _stream(img)), "mp_visualize")
    return out, mask


def prepare_inputs(segm, mean, std, with_color=True):
    """RTL/main.py:352-364 in one kernel: segm [1,4,H,W] (RGB in [-1,1] + mask) ->
    (input_netG [1,3,H,W] normalised and background-zeroed, input_netC [1,3,H,W] or None)."""
    sg = _f32c(segm)
    if sg.dim() != 4 or sg.shape[0] != 1 or sg.shape[1] != 4:
        raise ValueError("segm must be [1,4,H,W], got %s" % (tuple(segm.shape),))
    ctx = get_context(sg.device)
    hw = sg.shape[2] * sg.shape[3]
    g = torch.empty((1, 3) + tuple(sg.shape[2:]), dtype=torch.float32, device=sg.device)
    c = torch.empty_like(g) if with_color else None
    mean_c = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(mean, np.float32).reshape(3)])
    std_c = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(std, np.float32).reshape(3)])
    ctx.check(ctx.lib.mp_prepare_inputs(ctx.handle, _ptr(sg), hw, mean_c, std_c, _ptr(g),
                                        _ptr(c) if c is not None else None, _stream(sg)),
              "mp_prepare_inputs")
    return g, c


def marching_cubes_raw(volume, level=0.5, b_min=(-1, -1, -1), b_max=(1, 1, 1), max_verts=None,
                       max_faces=None):
    """mp_marching_cubes: capacity-sized (verts [max_v,3] f32, faces [max_f,3] int32,
    counts int32[2] = needed vertices / faces) on device, no host sync."""
    vol = volume
    while vol.dim() > 3:
        vol = vol[0]
    vol = _f32c(vol)
    r = vol.shape[0]
    ctx = get_context(vol.device)
    if max_verts is None:
        max_verts = 12 * r * r  # a closed body at resolution r has O(r^2) surface cells
    if max_faces is None:
        max_faces = 2 * max_verts
    dev = vol.device
    verts = torch.empty((max_verts, 3), dtype=torch.float32, device=dev)
    faces = torch.empty((max_faces, 3), dtype=torch.int32, device=dev)
    counts = torch.empty((2,), dtype=torch.int32, device=dev)
    bmin = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(b_min, np.float32).reshape(3)])
    bmax = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(b_max, np.float32).reshape(3)])
    ctx.check(ctx.lib.mp_marching_cubes(ctx.handle, _ptr(vol), r, float(level), bmin, bmax,
                                        _ptr(verts), max_verts, _ptr(faces), max_faces,
                                        _ptr(counts), _stream(vol)), "mp_marching_cubes")
    return verts, faces, counts


def group_norm(x, groups, weight, bias, eps=1e-5, relu=False):
    """[relu](GroupNorm(x)) for x [N,C,H,W] f32 contiguous on the GPU (mp_group_norm)."""
    ctx = get_encoder_context(x.device)
    n, c = x.shape[0], x.shape[1]
    hw = x.shape[2] * x.shape[3]
    y = torch.empty_like(x)
    ctx.check(ctx.lib.mp_group_norm(ctx.handle, _ptr(x), n, c, hw, int(groups), _ptr(weight),
                                    _ptr(bias), float(eps), int(bool(relu)), _ptr(y), _stream(x)),
              "mp_group_norm")
    return y


def group_norm_supported(x):
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32
            and x.is_contiguous() and (x.shape[2] * x.shape[3]) % 4 == 0)


def upsample_bicubic2x(x, add=None):
    """[add +] F.interpolate(x, scale_factor=2, mode='bicubic', align_corners=True), x [N,C,H,W]
    (the batch is folded into the channel axis: every plane is resampled independently)."""
    ctx = get_encoder_context(x.device)
    n, c, h, w = x.shape
    y = torch.empty((n, c, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    ctx.check(ctx.lib.mp_upsample_bicubic2x(ctx.handle, _ptr(x), n * c, h, w,
                                            _ptr(add) if add is not None else None, _ptr(y),
                                            _stream(x)), "mp_upsample_bicubic2x")
    return y


def concat3_add_supported(a, b, c, shortcut):
    ts = (a, b, c, shortcut)
    return (all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 for t in ts)
            and (a.shape[2] * a.shape[3]) % 4 == 0
            and a.shape[1] + b.shape[1] + c.shape[1] == shortcut.shape[1])


def concat3_add(a, b, c, shortcut):
    """torch.cat((a, b, c), 1) + shortcut in one pass (the tail of the encoders' ConvBlock)."""
    a, b, c, shortcut = _f32c(a), _f32c(b), _f32c(c), _f32c(shortcut)
    ctx = get_encoder_context(a.device)
    n, ca, h, w = a.shape
    y = torch.empty_like(shortcut)
    ctx.check(ctx.lib.mp_concat3_add(ctx.handle, _ptr(a), ca, _ptr(b), b.shape[1], _ptr(c),
                                     c.shape[1], _ptr(shortcut), n, h * w, _ptr(y), _stream(a)),
              "mp_concat3_add")
    return y


# MONOPORT_CONV_WINOGRAD=0: never pack Winograd-domain weights (every 3x3 convolution on the direct kernels)
CONV_WINOGRAD = os.environ.get("MONOPORT_CONV_WINOGRAD", "1") != "0"


class PackedConv3x3:
    """nn.Conv2d(Cin, Cout, 3, 1, 1, bias=False) weights in the MFMA fragment order of
    csrc/conv3x3.hip: exact f32 (mp_conv3x3_pack) or pre-split f16 halves for the "f16x3"
    arithmetic (mp_conv3x3_pack16; as accurate as f32, 5.3x fewer matrix cycles)."""

    def __init__(self, weight, precision="f32"):
        w = _f32c(weight.detach())
        self.cout, self.cin = int(w.shape[0]), int(w.shape[1])
        if tuple(w.shape[2:]) != (3, 3):
            raise ValueError("PackedConv3x3 wants a [Cout,Cin,3,3] weight, got %s" % (tuple(w.shape),))
        if precision not in ("f32", "f16x3"):
            raise ValueError("conv precision must be 'f32' or 'f16x3'")
        self.precision = precision
        ctx = get_encoder_context(w.device)
        self.data = torch.empty((w.numel(),), dtype=torch.float32, device=w.device)  # same bytes either way
        self.wmax = None
        self.wino = None  # Winograd-domain weights (csrc/conv_wino.hip) for the shapes that kernel serves
        if precision == "f32":
            ctx.check(ctx.lib.mp_conv3x3_pack(ctx.handle, _ptr(w), self.cout, self.cin, _ptr(self.data),
                                              _stream(w)), "mp_conv3x3_pack")
            if CONV_WINOGRAD and self.cout % 64 == 0:
                self.wino = torch.empty((16 * self.cout * self.cin,), dtype=torch.float32, device=w.device)
                ctx.check(ctx.lib.mp_conv3x3_pack_wino(ctx.handle, _ptr(w), self.cout, self.cin, _ptr(self.wino),
                                                       _stream(w)), "mp_conv3x3_pack_wino")
        else:
            self.wmax = torch.zeros((1,), dtype=torch.float32, device=w.device)
            ctx.check(ctx.lib.mp_conv3x3_pack16(ctx.handle, _ptr(w), self.cout, self.cin, _ptr(self.data),
                                                _ptr(self.wmax), _stream(w)), "mp_conv3x3_pack16")
        w.record_stream(torch.cuda.current_stream(w.device))


def conv3x3_supported(cin, cout, h, w):
    """True if csrc/conv3x3.hip is built for this shape (mp_conv3x3_supported)."""
    return bool(_lib.load().mp_conv3x3_supported(int(cin), int(cout), int(h), int(w)))


def scale_shift_add(t, ss, res):
    """res + (t * scale + shift): x + GroupNorm(t) with (scale, shift) from ``gn_finalize``."""
    ctx = get_encoder_context(t.device)
    t, res = t.contiguous(), res.contiguous()
    n, c = t.shape[0], t.shape[1]
    y = torch.empty_like(t)
    ctx.check(ctx.lib.mp_scale_shift_add(ctx.handle, _ptr(t), _ptr(ss), _ptr(res), n, c,
                                         t.shape[2] * t.shape[3], _ptr(y), _stream(t)), "mp_scale_shift_add")
    return y


def conv3x3_gn(x, ss, packed, relu=True, want_stats=False, reflect=False):
    """y = conv3x3(relu?(x * scale + shift)) (stride 1, zero padding 1 -- or ReflectionPad2d(1) with
    ``reflect`` -- no bias) as one MFMA kernel; ``ss`` [N,Cin,2] from ``gn_finalize`` or None (plain
    x).  Returns (y, stats) where
    stats = (partial sums double [N,32,S,2], S) of GroupNorm(32, Cout) over y, or None."""
    ctx = get_encoder_context(x.device)
    n, cin, h, w = x.shape
    if cin != packed.cin:
        raise ValueError("conv3x3_gn: input has %d channels, weights expect %d" % (cin, packed.cin))
    y = torch.empty((n, packed.cout, h, w), dtype=torch.float32, device=x.device)
    stats = None
    if want_stats:
        s = ctx.lib.mp_conv3x3_stat_slices(packed.cout, n, h, w, int(packed.precision == "f16x3"))
        stats = (torch.empty((n, 32, s, 2), dtype=torch.float64, device=x.device), s)
    if packed.precision == "f32":
        ctx.check(ctx.lib.mp_conv3x3_gn(ctx.handle, _ptr(x), n, cin, h, w,
                                        _ptr(ss) if ss is not None else None, int(bool(relu)),
                                        int(bool(reflect)), _ptr(packed.data), packed.cout, _ptr(y),
                                        _ptr(stats[0]) if stats else None, _stream(x)), "mp_conv3x3_gn")
    else:
        ctx.check(ctx.lib.mp_conv3x3_gn16(ctx.handle, _ptr(x), n, cin, h, w,
                                          _ptr(ss) if ss is not None else None, int(bool(relu)),
                                          int(bool(reflect)), _ptr(packed.data), _ptr(packed.wmax),
                                          packed.cout, _ptr(y),
                                          _ptr(stats[0]) if stats else None, _stream(x)),
                  "mp_conv3x3_gn16")
    return y, stats


class PackedConv1x1:
    """Weights of one fused 1x1 convolution in MFMA fragment order: W1 [Cout,C1(,1,1)] and optionally
    W2 [Cout,C2(,1,1)] (second K segment), biases summed (``b1`` may be None).  Cout = 256 (the
    hourglass tail) or 128 / 256 (the projection shortcut of a pyramid block)."""

    def __init__(self, w1, b1=None, w2=None, b2=None, precision="f32"):
        w1 = _f32c(w1.detach().reshape(w1.shape[0], -1))
        self.cout = int(w1.shape[0])
        if self.cout not in (128, 256):
            raise ValueError("conv1x1 is built for 128 or 256 output channels, got %d" % self.cout)
        self.c1, self.c2 = int(w1.shape[1]), 0
        if w2 is not None:
            w2 = _f32c(w2.detach().reshape(w2.shape[0], -1))
            self.c2 = int(w2.shape[1])
        self.precision = precision
        ctx = get_encoder_context(w1.device)
        self.data = torch.empty((self.cout * (self.c1 + self.c2),), dtype=torch.float32, device=w1.device)
        self.wmax = torch.zeros((1,), dtype=torch.float32, device=w1.device)
        bias = None if b1 is None else b1.detach().float()
        if b2 is not None:
            bias = b2.detach().float() if bias is None else bias + b2.detach().float()
        self.bias = None if bias is None else bias.contiguous()
        ctx.check(ctx.lib.mp_conv1x1_pack(ctx.handle, _ptr(w1), self.c1, _ptr(w2) if w2 is not None else None,
                                          self.c2, self.cout, int(precision == "f16x3"), _ptr(self.data),
                                          _ptr(self.wmax), _stream(w1)), "mp_conv1x1_pack")
        stream = torch.cuda.current_stream(w1.device)
        w1.record_stream(stream)
        if w2 is not None:
            w2.record_stream(stream)


def conv1x1_supported(x):
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[1] % 64 == 0
            and (x.shape[2] * x.shape[3]) % 64 == 0)


def conv1x1(x1, ss1, relu1, x2, packed, res=None, want_nchw=True, y_hwc=None, want_stats=False):
    """y = W [relu?(x1 * scale + shift) ; x2] + bias (+ res) as one fused GEMM (mp_conv1x1).
    Returns (y [N,Cout,H,W] or None, stats or None); ``y_hwc`` [N,H,W,256] is filled when given."""
    ctx = get_encoder_context(x1.device)
    x1 = x1.contiguous()
    n, c1, h, w = x1.shape
    hw = h * w
    if c1 != packed.c1 or (x2 is None) != (packed.c2 == 0) or (x2 is not None and x2.shape[1] != packed.c2):
        raise ValueError("conv1x1: inputs do not match the packed weights")
    if x2 is not None:
        x2 = x2.contiguous()
    if res is not None:
        res = res.contiguous()
    y = torch.empty((n, packed.cout, h, w), dtype=torch.float32, device=x1.device) if want_nchw else None
    stats = None
    if want_stats:
        s = ctx.lib.mp_conv1x1_stat_slices(hw)
        stats = (torch.empty((n, 32, s, 2), dtype=torch.float64, device=x1.device), s)
    if y_hwc is not None:
        assert y_hwc.is_contiguous() and y_hwc.numel() == n * hw * 256 and y_hwc.dtype == torch.float32
    ctx.check(ctx.lib.mp_conv1x1(
        ctx.handle, _ptr(x1), _ptr(ss1) if ss1 is not None else None, int(bool(relu1)),
        _ptr(x2) if x2 is not None else None, n, c1, packed.c2, packed.cout, hw, _ptr(packed.data),
        int(packed.precision == "f16x3"), _ptr(packed.wmax),
        _ptr(packed.bias) if packed.bias is not None else None,
        _ptr(res) if res is not None else None, _ptr(y) if y is not None else None,
        _ptr(y_hwc) if y_hwc is not None else None, _ptr(stats[0]) if stats else None, _stream(x1)),
        "mp_conv1x1")
    return y, stats


def gn_stats(x, groups):
    """One read pass over x [N,C,H,W]: (partial sums double [N*groups, S, 2], S)."""
    ctx = get_encoder_context(x.device)
    n, c = x.shape[0], x.shape[1]
    hw = x.shape[2] * x.shape[3]
    s = ctx.lib.mp_gn_stat_slices()
    partial = torch.empty((n * groups, s, 2), dtype=torch.float64, device=x.device)
    ctx.check(ctx.lib.mp_gn_stats(ctx.handle, _ptr(x), n, c, hw, int(groups), _ptr(partial),
                                  _stream(x)), "mp_gn_stats")
    return partial, s


def gn_finalize(stats, n, c, groups, count, weight, bias, eps):
    """Partial sums -> ss [N,C,2] = (gamma rstd, beta - mean gamma rstd) of GroupNorm(groups, C)."""
    partial, slices = stats
    ctx = get_encoder_context(partial.device)
    ss = torch.empty((n, c, 2), dtype=torch.float32, device=partial.device)
    ctx.check(ctx.lib.mp_gn_finalize(ctx.handle, _ptr(partial), n, c, int(groups), int(slices),
                                     int(count), _ptr(weight), _ptr(bias), float(eps), _ptr(ss),
                                     _stream(partial)), "mp_gn_finalize")
    return ss


# ---- GroupNorm hand-over from producer to consumer (include/monoport_hip.h, csrc/gn_tail.h) ----
def gn_acc_zeros(device, n, slots=None):
    """Zeroed GroupNorm accumulator(s): int64 [R,N,32,4] (or [slots,R,N,32,4]), R = mp_gn_acc_replicas()."""
    r = _lib.load().mp_gn_acc_replicas()
    shape = (r, n, 32, 4) if slots is None else (slots, r, n, 32, 4)
    return torch.zeros(shape, dtype=torch.int64, device=device)


class GnArena:
    """Accumulators [slots, R, N, 32, 4] int64 for the GroupNorms of one encoder pass, zeroed by ONE fill
    kernel; ``take()`` hands out the next [R,N,32,4] slice.  A producing kernel adds the statistics of
    the tensor it writes into its slice, the consuming kernel reads them (``gn=(acc, module)``)."""

    def __init__(self, device, n, slots):
        self.buf = gn_acc_zeros(device, n, slots)
        self.used = 0
        rec = _recording()
        if rec is not None:  # a replay clears the same arena again
            m = _lib.PlanMemsetArgs()
            m.ptr, m.bytes, m.value = self.buf.data_ptr(), self.buf.numel() * 8, 0
            rec.add(_lib.PLAN_MEMSET, m, [self.buf])

    def take(self):
        if self.used >= self.buf.shape[0]:
            raise RuntimeError("GnArena: more GroupNorms than slots (%d)" % self.buf.shape[0])
        acc = self.buf[self.used]
        self.used += 1
        return acc


def gn_reference_ss(acc, gn, count):
    """(scale, shift) [N,C,2] a consumer derives from an accumulator -- host-side restatement of
    gn_load_stats / gn_scale_shift for tests and probes (not used by the product path)."""
    a = acc.sum(0).to(torch.float64)  # the replicas add up as integers
    lo_s = torch.where(a[..., 1] < 0, a[..., 1] + 2.0 ** 64, a[..., 1])
    lo_q = torch.where(a[..., 3] < 0, a[..., 3] + 2.0 ** 64, a[..., 3])
    s = a[..., 0] / 65536.0 + lo_s / 2.0 ** 64
    q = a[..., 2] / 65536.0 + lo_q / 2.0 ** 64
    mean = s / count
    var = torch.clamp(q / count - mean * mean, min=0.0)
    rstd = (1.0 / torch.sqrt(var + gn.eps)).float()
    mean = mean.float()
    cpg = gn.num_channels // 32
    sc = rstd.repeat_interleave(cpg, 1) * gn.weight[None]
    sh = gn.bias[None] - mean.repeat_interleave(cpg, 1) * sc
    return torch.stack((sc, sh), 2)


def _gn_in(dst, gn, c):
    """Fill a _lib.GnIn: ``gn`` = None (plain input), (acc [N,32,4] int64, GroupNorm module) or a
    precomputed ss [N,C,2] tensor (legacy, from gn_finalize)."""
    if gn is None:
        return
    if torch.is_tensor(gn):
        dst.ss = gn.data_ptr()
        return
    acc, mod = gn
    if mod.num_groups != 32 or mod.num_channels != c:
        raise ValueError("GroupNorm(%d, %d) does not match a %d-channel input" % (mod.num_groups, mod.num_channels, c))
    dst.acc = acc.data_ptr()
    dst.gamma = mod.weight.data_ptr()
    dst.beta = mod.bias.data_ptr()
    dst.eps = float(mod.eps)


def _gn_keep(gn):
    """Tensors a recorded command's GroupNorm input points at (accumulator; the module owns gamma / beta)."""
    if gn is None:
        return []
    if torch.is_tensor(gn):
        return [gn]
    return [gn[0], gn[1].weight, gn[1].bias]


def _gn_out(dst, acc):
    if acc is not None:
        assert acc.dtype == torch.int64 and acc.is_contiguous()
        dst.acc = acc.data_ptr()


def conv3x3_fused(x, gn, packed, relu=True, reflect=False, want_y=True, stats=None, out=None, res=None,
                  out_off=0, out_stats=None):
    """mp_conv3x3_ex: y = conv3x3(relu?(GroupNorm(x))) with the GroupNorm hand-over and, optionally, the
    pyramid block's tail fused into the epilogue.
      gn         GroupNorm of the input: None, (acc, module) or a legacy ss tensor
      stats      accumulator [R,N,32,4] that receives the statistics of y (for the GroupNorm reading y)
      out, res   [N,Ctot,H,W]: out[:, out_off:out_off+Cout] = y + res[:, same]  (cat + residual)
      out_stats  accumulator for GroupNorm(32, Ctot) over ``out`` (shared by the launches filling it)
    Returns y (or None with want_y=False)."""
    ctx = get_encoder_context(x.device)
    n, cin, h, w = x.shape
    if cin != packed.cin:
        raise ValueError("conv3x3: input has %d channels, weights expect %d" % (cin, packed.cin))
    a = _lib.Conv3x3Args()
    y = torch.empty((n, packed.cout, h, w), dtype=torch.float32, device=x.device) if want_y else None
    f16 = packed.precision == "f16x3"
    _gn_in(a.gn, gn, cin)
    _gn_out(a.fin, stats)
    if out is not None:
        assert out.is_contiguous() and res.is_contiguous() and out.shape == res.shape
        _gn_out(a.fin2, out_stats)
        a.y2, a.res = out.data_ptr(), res.data_ptr()
        a.y2_channels, a.y2_offset = out.shape[1], int(out_off)
    a.x, a.n, a.cin, a.h, a.w = x.data_ptr(), n, cin, h, w
    a.relu, a.reflect = int(bool(relu)), int(bool(reflect))
    a.packed = packed.data.data_ptr()
    a.wmax = packed.wmax.data_ptr() if f16 else None
    a.packed_wino = packed.wino.data_ptr() if packed.wino is not None else None
    a.cout = packed.cout
    a.y = y.data_ptr() if y is not None else None
    ctx.check(ctx.lib.mp_conv3x3_ex(ctx.handle, ctypes.byref(a), _stream(x)), "mp_conv3x3_ex")
    rec = _recording()
    if rec is not None:
        rec.add(_lib.PLAN_CONV3X3, a, [x, packed.data, packed.wmax, packed.wino, y, out, res, stats, out_stats] + _gn_keep(gn))
    return y


def conv1x1_fused(x1, gn1, relu1, x2, packed, res=None, want_nchw=True, y_hwc=None, stats=None):
    """mp_conv1x1_ex: ``conv1x1`` with the GroupNorm hand-over (gn1 / stats as in conv3x3_fused).
    Returns y (or None)."""
    ctx = get_encoder_context(x1.device)
    x1 = x1.contiguous()
    n, c1, h, w = x1.shape
    hw = h * w
    if c1 != packed.c1 or (x2 is None) != (packed.c2 == 0) or (x2 is not None and x2.shape[1] != packed.c2):
        raise ValueError("conv1x1: inputs do not match the packed weights")
    a = _lib.Conv1x1Args()
    y = torch.empty((n, packed.cout, h, w), dtype=torch.float32, device=x1.device) if want_nchw else None
    _gn_in(a.gn1, gn1, c1)
    _gn_out(a.fin, stats)
    if y_hwc is not None:
        assert y_hwc.is_contiguous() and y_hwc.numel() == n * hw * 256 and y_hwc.dtype == torch.float32
    a.x1 = x1.data_ptr()
    a.relu1 = int(bool(relu1))
    if x2 is not None:
        x2 = x2.contiguous()
        a.x2 = x2.data_ptr()
    if res is not None:
        res = res.contiguous()
        a.res = res.data_ptr()
    a.n, a.c1, a.c2, a.cout, a.hw = n, c1, packed.c2, packed.cout, hw
    a.packed = packed.data.data_ptr()
    a.f16 = int(packed.precision == "f16x3")
    a.wmax = packed.wmax.data_ptr()
    a.bias = packed.bias.data_ptr() if packed.bias is not None else None
    a.y = y.data_ptr() if y is not None else None
    a.y_hwc = y_hwc.data_ptr() if y_hwc is not None else None
    ctx.check(ctx.lib.mp_conv1x1_ex(ctx.handle, ctypes.byref(a), _stream(x1)), "mp_conv1x1_ex")
    rec = _recording()
    if rec is not None:
        rec.add(_lib.PLAN_CONV1X1, a, [x1, x2, res, packed.data, packed.wmax, packed.bias, y, y_hwc, stats] + _gn_keep(gn1))
    return y


class PackedConvK:
    """Weights of a 7x7 (3 -> 64) or 3x3 stride-2 convolution in the fragment order of
    csrc/convim2col.hip (mp_convk_pack); ``bias`` may be None."""

    def __init__(self, weight, bias=None):
        w = _f32c(weight.detach())
        self.cout, self.cin, self.ks = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
        ctx = get_encoder_context(w.device)
        n = ctx.lib.mp_convk_packed_floats(self.cin, self.cout, self.ks)
        if n <= 0 or w.shape[2] != w.shape[3]:
            raise ValueError("PackedConvK: unsupported weight %s" % (tuple(w.shape),))
        self.data = torch.empty((n,), dtype=torch.float32, device=w.device)
        self.bias = None if bias is None else _f32c(bias.detach())
        ctx.check(ctx.lib.mp_convk_pack(ctx.handle, _ptr(w), self.cout, self.cin, self.ks, _ptr(self.data),
                                        _stream(w)), "mp_convk_pack")
        w.record_stream(torch.cuda.current_stream(w.device))


def convk_supported(cin, cout, ks, stride, h, w):
    return bool(_lib.load().mp_convk_supported(int(cin), int(cout), int(ks), int(stride), int(h), int(w)))


def convk(x, gn, relu, packed, stride, reflect=False, stats=None):
    """mp_convk: y = conv_ks(relu?(GroupNorm(x))) (+ bias), stride 1 / 2, padding ks // 2 (zero or
    reflect); gn / stats as in conv3x3_fused.  Returns y."""
    ctx = get_encoder_context(x.device)
    x = x.contiguous()
    n, cin, h, w = x.shape
    a = _lib.ConvKArgs()
    y = torch.empty((n, packed.cout, h // stride, w // stride), dtype=torch.float32, device=x.device)
    _gn_in(a.gn, gn, cin)
    _gn_out(a.fin, stats)
    a.x, a.n, a.cin, a.h, a.w = x.data_ptr(), n, cin, h, w
    a.relu, a.reflect = int(bool(relu)), int(bool(reflect))
    a.packed = packed.data.data_ptr()
    a.bias = packed.bias.data_ptr() if packed.bias is not None else None
    a.cout, a.ks, a.stride = packed.cout, packed.ks, int(stride)
    a.y = y.data_ptr()
    ctx.check(ctx.lib.mp_convk(ctx.handle, ctypes.byref(a), _stream(x)), "mp_convk")
    rec = _recording()
    if rec is not None:
        rec.add(_lib.PLAN_CONVK, a, [x, packed.data, packed.bias, y, stats] + _gn_keep(gn))
    return y


def avgpool2_gn(x, stats=None):
    """F.avg_pool2d(x, 2, stride=2), adding the statistics of the result into ``stats``."""
    ctx = get_encoder_context(x.device)
    x = x.contiguous()
    n, c, h, w = x.shape
    y = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    fin = _lib.GnOut()
    _gn_out(fin, stats)
    ctx.check(ctx.lib.mp_avgpool2_gn(ctx.handle, _ptr(x), n, c, h, w, _ptr(y), ctypes.byref(fin), _stream(x)),
              "mp_avgpool2_gn")
    rec = _recording()
    if rec is not None:
        a = _lib.PlanPoolArgs()
        a.x, a.n, a.c, a.h, a.w, a.y, a.fin = x.data_ptr(), n, c, h, w, y.data_ptr(), fin
        rec.add(_lib.PLAN_AVGPOOL2, a, [x, y, stats])
    return y


def upsample_add_gn(x, add, stats=None):
    """add + bicubic x2 of x (HGFilters.py:108-111), adding the statistics of the result into ``stats``."""
    ctx = get_encoder_context(x.device)
    x = x.contiguous()
    n, c, h, w = x.shape
    y = torch.empty((n, c, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    if add is not None:
        add = add.contiguous()
    fin = _lib.GnOut()
    _gn_out(fin, stats)
    ctx.check(ctx.lib.mp_upsample_bicubic2x_gn(ctx.handle, _ptr(x), n, c, h, w,
                                               _ptr(add) if add is not None else None, _ptr(y),
                                               ctypes.byref(fin), _stream(x)), "mp_upsample_bicubic2x_gn")
    rec = _recording()
    if rec is not None:
        a = _lib.PlanUpsampleArgs()
        a.x, a.n, a.c, a.h, a.w, a.y, a.fin = x.data_ptr(), n, c, h, w, y.data_ptr(), fin
        a.add = add.data_ptr() if add is not None else None
        rec.add(_lib.PLAN_UPSAMPLE2X, a, [x, add, y, stats])
    return y


def gn_apply(x, gn, relu=True, res=None, stats=None):
    """[res +] relu?(GroupNorm(x)) materialised (gn = (acc, module) or a legacy ss tensor), adding the
    statistics of the result into ``stats``."""
    ctx = get_encoder_context(x.device)
    x = x.contiguous()
    n, c = x.shape[0], x.shape[1]
    hw = x.shape[2] * x.shape[3]
    y = torch.empty_like(x)
    g = _lib.GnIn()
    _gn_in(g, gn, c)
    fin = _lib.GnOut()
    _gn_out(fin, stats)
    if res is not None:
        res = res.contiguous()
    ctx.check(ctx.lib.mp_gn_apply(ctx.handle, _ptr(x), ctypes.byref(g), int(bool(relu)), n, c, hw,
                                  _ptr(res) if res is not None else None, _ptr(y), ctypes.byref(fin),
                                  _stream(x)), "mp_gn_apply")
    rec = _recording()
    if rec is not None:
        a = _lib.PlanGnApplyArgs()
        a.x, a.gn, a.relu, a.n, a.c, a.hw, a.y, a.fin = x.data_ptr(), g, int(bool(relu)), n, c, hw, y.data_ptr(), fin
        a.res = res.data_ptr() if res is not None else None
        rec.add(_lib.PLAN_GN_APPLY, a, [x, res, y, stats] + _gn_keep(gn))
    return y


def memory_stats(device):
    """mp_memory_stats: device memory the contexts of ``device`` own (scratch arenas incl. outgrown blocks, packed MLP
    weights) and how many arenas / registered skip tables it tracks."""
    tot = [0, 0, 0, 0]
    for ctx in (get_context(device), get_encoder_context(device)):
        out = (ctypes.c_int64 * 4)()
        ctx.check(ctx.lib.mp_memory_stats(ctx.handle, out), "mp_memory_stats")
        tot = [a + int(b) for a, b in zip(tot, out)]
    return {"arena_bytes": tot[0], "weight_bytes": tot[1], "arenas": tot[2], "skip_tables": tot[3]}


def mfma_clock_probe(device, ms_target=20.0, stream=None):
    """What the f32 matrix pipe of ``device`` sustains right now (mp_mfma_clock_probe, csrc/clock_probe.hip):
    {"tflops", "shader_clock_mhz", "ms", "workgroups"} of a register-only MFMA loop of about ``ms_target`` ms."""
    ctx = get_context(device)
    out = (ctypes.c_double * 4)()
    st = stream if stream is not None else torch.cuda.current_stream(torch.device(device))
    ctx.check(ctx.lib.mp_mfma_clock_probe(ctx.handle, float(ms_target), out, ctypes.c_void_p(st.cuda_stream)),
              "mp_mfma_clock_probe")
    return {"tflops": out[0], "shader_clock_mhz": out[1], "ms": out[2], "workgroups": int(out[3])}


def profile_begin(device, max_records=4096):
    """Start bracketing fused-query launches on ``device`` with HIP events (bench.py roofline)."""
    ctx = get_context(device)
    ctx.check(ctx.lib.mp_profile_begin(ctx.handle, int(max_records)), "mp_profile_begin")


def profile_end(device, capacity=4096):
    """Stop and return the per-launch durations (ms, launch order) as a numpy array."""
    ctx = get_context(device)
    buf = (ctypes.c_float * capacity)()
    n = ctypes.c_int(0)
    ctx.check(ctx.lib.mp_profile_end(ctx.handle, buf, capacity, ctypes.byref(n)), "mp_profile_end")
    return np.array(buf[:min(n.value, capacity)], dtype=np.float64)
