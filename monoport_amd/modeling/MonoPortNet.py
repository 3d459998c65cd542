"""MonoPortNet: the PIFu geometry / colour network wrapper (mirror of
monoport/lib/modeling/MonoPortNet.py, same constructor, attributes and method signatures).

``filter`` runs the image encoder once per frame as a chain of hand-written HIP kernels
(modeling/backbones.py over csrc/conv3x3.hip, convim2col.hip, encoder_ops.hip; torch ops only in
train mode / on CPU tensors).  ``query`` -- the hot path, called once per octree level on
10^4..10^5 points -- is one fused HIP kernel launch: projection, in-image mask, depth feature,
bilinear gather, the skip-connected MLP on f32 MFMA, final activation and mask.  Which kernel:
csrc/query_table.hip when the bound feature map has a skip table (``_skip_table`` below: the
octree engine's maps, and maps that have served 16 k points), else csrc/query.hip /
query_small.hip; netC (C = 512) always csrc/query.hip.
"""
import collections
import threading
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .backbones import PIFuHGFilters, PIFuResBlkFilters
from .geometry import index, orthogonal, perspective  # noqa: F401
from .heads import PIFuNetCMLP, PIFuNetGMLP
from .normalizers import PIFuNomalizer

# the reference resolves these names through globals() (MonoPortNet.py:23-28)
_REGISTRY = {
    "PIFuHGFilters": PIFuHGFilters,
    "PIFuResBlkFilters": PIFuResBlkFilters,
    "PIFuNetGMLP": PIFuNetGMLP,
    "PIFuNetCMLP": PIFuNetCMLP,
    "PIFuNomalizer": PIFuNomalizer,
    "orthogonal": orthogonal,
    "perspective": perspective,
}

_tls = threading.local()


class QueryBinding:
    """What one ``MonoPortNet.query`` call binds together: packed MLP + channels-last features +
    calibration.  The octree engine records it once per frame and drives all levels natively."""

    def __init__(self, net, mlp, feat_hwc, calib, z_scale):
        self.net, self.mlp, self.feat_hwc, self.calib, self.z_scale = net, mlp, feat_hwc, calib, z_scale


class record_query:
    """Context manager (per host thread): counts the ``MonoPortNet.query`` calls made inside it
    and keeps the QueryBinding of the first one in ``.binding``.  The calls run normally.  With
    ``capture_only=True`` the first call returns zeros instead of launching (a probe).  Used by
    Seg3dLossless to see through an opaque ``query_func`` closure such as RTL/main.py:169-183."""

    def __init__(self, capture_only=False):
        self.capture_only = capture_only

    def __enter__(self):
        self.binding = None
        self.calls = 0
        self._prev = getattr(_tls, "capture", None)
        _tls.capture = self
        return self

    def __exit__(self, *exc):
        _tls.capture = self._prev
        return False


def capture_query():
    """Probe form of ``record_query`` (the first query call is recorded, not executed)."""
    return record_query(capture_only=True)


class _AttrDict(dict):
    """yacs-free stand-in for CfgNode in the factories below."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class MonoPortNet(nn.Module):
    def __init__(self, opt_net):
        super().__init__()
        self.opt = opt_net
        assert opt_net.projection in ["orthogonal", "perspective"]
        self.image_filter = _REGISTRY[opt_net.backbone.IMF](opt_net.backbone)
        self.surface_classifier = _REGISTRY[opt_net.head.IMF](opt_net.head)
        self.projection = _REGISTRY[opt_net.projection]
        self.normalizer = _REGISTRY[opt_net.normalizer.IMF](opt_net.normalizer)
        # per bound feature map, most recent last (a stage pipeline keeps several frames in flight, and a
        # coalescing recon stage binds up to 16 of them before it launches): key -> (weakrefs of the
        # source maps, packed channels-last map); id(packed) -> [packed, its version, mlp, skip table or
        # None, mlp generation, query points served so far]
        self._hwc_cache = collections.OrderedDict()
        self._table_cache = collections.OrderedDict()

    # ---- encoder ---------------------------------------------------------------------------------
    def filter(self, images, feat_prior=None):
        """images [B,3,512,512] -> list(stages) of list(levels) of [B,C,128,128]
        (MonoPortNet.py:31-46).  With ``feat_prior`` (netC) the prior is nearest-resized to
        128x128 and concatenated FIRST (:42-44)."""
        feats_stages = self.image_filter(images)
        if feat_prior is not None:
            feat_prior = F.interpolate(feat_prior, size=(128, 128))
            feats_stages = [[torch.cat([feat_prior, f], dim=1) for f in feats]
                            for feats in feats_stages]
        return feats_stages

    # ---- hot path --------------------------------------------------------------------------------
    MAX_BOUND_MAPS = ops.MAX_FRAMES  # packed maps / skip tables kept at most (a coalescing stage binds up to kMaxFrames)

    def _drop_dead_maps(self):
        """Forget the packed copy and the skip table (128 MB) of every feature map whose source tensors
        have been freed: the reference's one-frame-at-a-time usage then holds ONE map, a pipeline with k
        frames in flight k of them (the LRU bound only caps a caller that keeps every map alive)."""
        for key in [k for k, (refs, _) in self._hwc_cache.items() if any(r() is None for r in refs)]:
            _, packed = self._hwc_cache.pop(key)
            e = self._table_cache.pop(id(packed), None)
            if e is not None and e[3] is not None:
                e[3].release()

    def _packed_features(self, feats):
        """Channels-last copy of this stage's maps, cached per source tensors so the five octree
        levels of one frame (and repeated calls) pack once."""
        key = tuple((f.data_ptr(), f._version, tuple(f.shape)) for f in feats)
        c = self._hwc_cache.get(key)
        if c is not None and all(r() is f for r, f in zip(c[0], feats)):
            self._hwc_cache.move_to_end(key)
            return c[1]
        self._drop_dead_maps()
        packed = ops.pack_features(list(feats))
        self._hwc_cache[key] = ([weakref.ref(f) for f in feats], packed)
        while len(self._hwc_cache) > self.MAX_BOUND_MAPS:
            self._hwc_cache.popitem(last=False)
        return packed

    def bind(self, feats_stages, calibs, n_points=0, for_engine=False):
        """QueryBinding for eval-mode queries against ``feats_stages`` / ``calibs``.
        ``n_points``: how many points the caller is about to query; ``for_engine``: the caller is
        the octree engine (a whole reconstruction follows) -- both feed the decision whether the
        map gets a skip table (``_skip_table``)."""
        if self.training:
            raise NotImplementedError("monoport_amd implements the inference path (net.eval())")
        if self.projection is not orthogonal:
            raise NotImplementedError("only the orthogonal projection of the PIFu configs is built")
        feats = list(feats_stages[-1])  # eval keeps the last stage only (MonoPortNet.py:63-64)
        dev = feats[0].device
        if calibs is None:
            calibs = torch.eye(4, device=dev)[None]  # xyz = points (MonoPortNet.py:66-67)
        mlp = self.surface_classifier.packed()
        if mlp.ctx.device_index != (dev.index if dev.index is not None else torch.cuda.current_device()):
            raise RuntimeError("surface_classifier and the feature maps must be on one GPU "
                               "(RTL/main.py:382-387 moves the features first)")
        packed = self._packed_features(feats)
        self._skip_table(mlp, packed, int(n_points), for_engine)
        return QueryBinding(self, mlp, packed, calibs, self.normalizer.scale)

    def _skip_table(self, mlp, packed, n_points=0, for_engine=True):
        """The skip table of the bound feature map (ops.skip_table: the MLP's products with the
        sampled feature, taken once per texel instead of once per query point), registered for the
        map so that every query of the frame -- this module's and the octree engine's -- blends
        table rows.  A table costs 16 GFLOP / 126 MB whatever follows, the work of ~16 k plain-path
        points: it is made when the octree engine binds the map (a reconstruction of ~3e5 points
        follows) or once the map has served ops.SKIP_TABLE_MIN_POINTS query points; a few small
        ``query`` calls stay on the plain kernels (the two paths differ by f32 rounding, 1-5e-7).
        netG heads (C = 256) whose precision the C side routes through tables (ops.table_precision: exact
        f32 and f16x3; f16w / f16 measured slower through them and stay on the plain kernel, so no table
        is built for them); the table itself is always exact f32; MONOPORT_SKIP_TABLE=off (ops.SKIP_TABLE)
        switches it off.  The tables of live bound maps stay registered, at most MAX_BOUND_MAPS.
        Determinism: for plain ``query`` calls the threshold is CUMULATIVE per map, so the same call can
        return results that differ in the last bits (<= 5e-7, the distance between the two kernels) before
        and after the map has earned its table; MONOPORT_SKIP_TABLE_MIN_POINTS=0 (always) or
        MONOPORT_SKIP_TABLE=off (never) make every call take the same path."""
        cache = self._table_cache
        for k in [k for k, e in cache.items()  # entries of recycled / rewritten maps or of other weights
                  if e[0]._version != e[1] or e[2] is not mlp or e[4] != mlp.generation or not ops.SKIP_TABLE]:
            e = cache.pop(k)
            if e[3] is not None:
                e[3].release()
        e = cache.get(id(packed))
        if e is None or e[0] is not packed:
            e = cache[id(packed)] = [packed, packed._version, mlp, None, mlp.generation, 0]
        cache.move_to_end(id(packed))
        e[5] += n_points
        h, w, ch = packed.shape
        wanted = ops.SKIP_TABLE and ch == 256 and (h * w) % 64 == 0 and ops.table_precision(mlp.precision)
        if wanted and e[3] is None and (for_engine or e[5] >= ops.SKIP_TABLE_MIN_POINTS):
            # the handle keeps map and table alive and unregisters them when it is dropped
            e[3] = ops.skip_table(mlp, packed)
        while len(cache) > self.MAX_BOUND_MAPS:
            _, old = cache.popitem(last=False)
            if old[3] is not None:
                old[3].release()

    def has_skip_table(self, packed=None):
        """Whether the most recently bound map (or ``packed``) has a registered skip table."""
        if not self._table_cache:
            return False
        e = self._table_cache.get(id(packed)) if packed is not None else next(reversed(self._table_cache.values()))
        return e is not None and e[3] is not None

    def query(self, feats_stages, points, calibs=None, transforms=None):
        """points [B,3,N] world coords -> [ [B,Cout,N] ] (MonoPortNet.py:48-91, eval mode).
        Out-of-image points come back as exactly 0 (:89)."""
        if transforms is not None:
            raise NotImplementedError("query(transforms=...) is a training-time option")
        cap = getattr(_tls, "capture", None)
        binding = self.bind(feats_stages, calibs, n_points=points.shape[2], for_engine=cap is not None)
        if cap is not None:
            cap.calls += 1
            if cap.binding is None:
                cap.binding = binding
                if cap.capture_only:
                    return [torch.zeros((points.shape[0], binding.mlp.cout, points.shape[2]),
                                        dtype=torch.float32, device=points.device)]
        if points.shape[0] != 1:
            raise NotImplementedError("batch size 1 (RTL/main.py:175 asserts the same)")
        return [ops.query(binding.mlp, binding.feat_hwc, points, binding.calib, binding.z_scale)]

    def get_loss(self, pred_stages, labels):
        """Average MSE / L1 over stages (MonoPortNet.py:93-117); plain tensor ops."""
        kind = self.opt.loss.IMF
        if kind not in ("MSE", "L1"):
            raise NotImplementedError(kind)
        fn = F.mse_loss if kind == "MSE" else F.l1_loss
        return sum(fn(p, labels) for p in pred_stages) / len(pred_stages)

    def forward(self, images, points, calibs, transforms=None, labels=None, feat_prior=None):
        feats_stages = self.filter(images, feat_prior)
        pred_stages = self.query(feats_stages, points, calibs, transforms)
        if labels is not None:
            return pred_stages[-1], self.get_loss(pred_stages, labels)
        return pred_stages[-1]

    def load_legacy_pifu(self, ckpt_path):
        """Flat legacy PIFu checkpoints: ``image_filter.*`` and ``surface_classifier.conv{i}.*``
        (renamed to ``filters.{i}.*``) -- MonoPortNet.py:153-160."""
        ckpt = torch.load(ckpt_path, map_location="cpu")
        self.image_filter.load_state_dict(
            {k.replace("image_filter.", ""): v for k, v in ckpt.items() if "image_filter" in k})
        self.surface_classifier.load_state_dict(
            {k.replace("surface_classifier.conv", "filters."): v
             for k, v in ckpt.items() if "surface_classifier" in k})


def _options(backbone, head, loss):
    opt = _AttrDict(projection="orthogonal")
    opt.backbone = _AttrDict(IMF=backbone)
    opt.normalizer = _AttrDict(IMF="PIFuNomalizer")
    opt.head = _AttrDict(IMF=head)
    opt.loss = _AttrDict(IMF=loss)
    return opt


def PIFuNetG():
    """netG: hourglass encoder + [257,1024,512,256,128,1] sigmoid head (MonoPortNet.py:163-184)."""
    return MonoPortNet(_options("PIFuHGFilters", "PIFuNetGMLP", "MSE"))


def PIFuNetC():
    """netC: ResNet encoder + [513,1024,512,256,128,3] tanh head (MonoPortNet.py:187-208)."""
    return MonoPortNet(_options("PIFuResBlkFilters", "PIFuNetCMLP", "L1"))
