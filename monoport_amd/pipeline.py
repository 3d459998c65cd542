"""Per-frame reconstruction pipeline with several frames in flight on one GPU.

The reference overlaps frames with one Python thread per stage (RTL/dataloader.py:734-751,
:1026-1053).  On an MI355X the stages are so short (tens of microseconds to a few milliseconds)
that host-side launch latency and GPU fill, not thread parallelism, are what matter:

* a ``FrameSlot`` owns a HIP stream and the static buffers of ``batch`` frames.  The image
  encoder -- 137 launches of the hand-written kernels of csrc/conv3x3.hip, convim2col.hip and
  encoder_ops.hip, no MIOpen / torch op -- runs ONCE per slot on the whole batch (a batch-1
  convolution on a 32^2 / 64^2 map under-fills 256 CUs: 3.8 ms/frame at batch 1, 2.1 ms/frame at
  batch 16) and is replayed as a hipGraph; the slot's skip tables (csrc/query_table.hip) are made
  in one launch behind it; then each frame goes through

      channels-last pack -> octree (5 levels, fused query) -> forward_vertices -> render

  as asynchronous C-ABI calls on the slot's stream, the octree level by level for the whole
  batch (``mp_recon_batch``: one fused-query launch per level covers all frames of the slot --
  a single frame's coarse levels, 5-25 k points, leave most of the 256 CUs idle);
* a ``FramePipeline`` round-robins over ``depth`` slots, so independent frames on different
  streams fill the CUs that the coarse octree levels and the small encoder kernels leave idle
  (``depth`` x ``batch`` frames in flight; BASELINE configs[3] asks for 8).
"""
import numpy as np
import torch

from . import ops
from .synthetic import Z_SCALE

import os

RESOLUTIONS = (17, 33, 65, 129, 257)  # RTL/main.py:187
# frames per forward_vertices / paint launch set (results are identical): "on" = all frames of the slot, "off" = one,
# or a number.  4: measured on 20-frame single submissions 185.8 (1) / 189.2 (4) / 189.8 (all) recon/s -- and with three
# slots overlapping, launches over all 16 frames of a slot (first_hit: 37 k workgroups) hold up the other slots'
# chains of small dependent kernels: passes of 5.4-5.6 ms per frame against 5.2-5.4 with 1 or 4 frames per launch
_vb = os.environ.get("MONOPORT_VERTEX_BATCH", "4")
VERTEX_BATCH = 0 if _vb == "off" else (int(_vb) if _vb.isdigit() else 10 ** 6)
MAX_RECON_BATCH = ops.MAX_FRAMES  # kMaxFrames of the C-ABI (include/monoport_hip.h, mp_recon_batch): 32


class FrameSlot:
    """Static buffers for ``batch`` in-flight frames (geometry chain of RTL/main.py:366-428, plus
    the netC texture stages :373-441 when ``netC`` is given)."""

    def __init__(self, netG, device, resolutions=RESOLUTIONS, b_min=(-1, -1, -1), b_max=(1, 1, 1),
                 balance=0.5, feature_hook=None, use_graph=False, netC=None, batch=1, skip_table=None,
                 final_level="dilate3"):
        self.net = netG
        self.netC = netC
        self.device = torch.device(device)
        self.res = [int(r) for r in resolutions]
        self.b_min, self.b_max, self.balance = b_min, b_max, float(balance)
        self.final_level = final_level  # the last octree level's selection rule (Seg3dLossless docstring)
        self.feature_hook = feature_hook  # optional in-place edit of the [B,C,H,W] feature map
        self.batch = int(batch)
        r = self.res[-1]
        dev = self.device
        b = self.batch
        self.stream = torch.cuda.Stream(device=dev)
        self.image = torch.zeros((b, 3, 512, 512), dtype=torch.float32, device=dev)
        self.calib = torch.eye(4, dtype=torch.float32, device=dev)[None].repeat(b, 1, 1).contiguous()
        # one [B,128,128,256] channels-last map; feats_hwc[b] are its per-frame views
        self.feat_hwc_all = torch.empty((b, 128, 128, 256), dtype=torch.float32, device=dev)
        self.feats_hwc = [self.feat_hwc_all[i] for i in range(b)]
        # skip tables of the slot's maps (ops.SKIP_TABLE unless the caller says otherwise) -- only for the MLP
        # precisions whose query kernel reads them (f32, f16x3): a table nobody reads is 16 GFLOP + 128 MB a frame
        self.skip_table = ((ops.SKIP_TABLE if skip_table is None else bool(skip_table))
                           and ops.table_precision(netG.surface_classifier.precision))
        self.tables = (torch.empty((b, 128, 128, ops.SKIP_TABLE_ROWS), dtype=torch.float32, device=dev)
                       if self.skip_table else None)
        self._table_handle = None  # registration of the tables of the last encoder pass
        # the hourglass encoder can write its last stack's features straight into that map
        # (HGFilter.forward(hwc_out=...), csrc/conv3x3.hip: conv1x1_kernel); a feature_hook must
        # then come with a channels-last twin, ``feature_hook.hwc(feat_hwc_all)``
        from .modeling import backbones
        self.hwc_direct = (isinstance(netG.image_filter, backbones.HGFilter)
                           and backbones.ENCODER_CONV == "hip"
                           and (feature_hook is None or hasattr(feature_hook, "hwc")))
        # one volume per frame of the batch: results stay readable until the next submit
        self.volumes = [torch.empty((r, r, r), dtype=torch.float32, device=dev) for _ in range(b)]
        self.status = torch.zeros((b, 1 + len(self.res)), dtype=torch.int32, device=dev)
        self.renders = [None] * b
        self.renders_tex = [None] * b
        self.vertices = [None] * b
        self.graph = None
        self._graph_feat = None
        self.use_graph = use_graph
        self._busy = False
        self.n_active = b  # frames of the current submission (a stream's last batch may be short)
        if netC is not None:
            from .recon import color_matrix
            # one channels-last netC map per frame: the colour queries of the whole slot are one launch
            self.feats_hwc_c = [torch.empty((128, 128, 512), dtype=torch.float32, device=dev)
                                for _ in range(b)]
            self.mat_color = color_matrix(b_min, b_max, r)
            self.image_c = torch.zeros((b, 3, 512, 512), dtype=torch.float32, device=dev)

    # convenience views for batch == 1 callers
    @property
    def volume(self):
        return self.volumes[0]

    @property
    def render(self):
        return self.renders[0]

    @property
    def render_tex(self):
        return self.renders_tex[0]

    @torch.no_grad()
    def _encode(self):
        """Both encoders on the slot's image buffers: (featG [B,256,128,128], featC or None)."""
        if self.hwc_direct:
            feat = self.net.image_filter(self.image, last_only=True, hwc_out=self.feat_hwc_all,
                                         keep_nchw=self.netC is not None, graphed=False)[-1][0]
            if self.feature_hook is not None:
                self.feature_hook.hwc(self.feat_hwc_all)
                if feat is not None:
                    self.feature_hook(feat)
        else:
            feat = self.net.image_filter(self.image, last_only=True)[-1][0]
            if self.feature_hook is not None:
                self.feature_hook(feat)
        feat_c = None
        if self.netC is not None:
            feat_c = self.netC.image_filter(self.image_c)[-1][0]  # [B,256,128,128]
        return feat, feat_c

    @torch.no_grad()
    def _chain(self):
        mlp = self.net.surface_classifier.packed()
        if self.graph is not None:
            self.graph.replay()  # the batched encoder(s) as one hipGraph launch
            feat, feat_c = self._graph_feat
        else:
            feat, feat_c = self._encode()
        if self.netC is not None:
            mlp_c = self.netC.surface_classifier.packed()
        r = self.res[-1]
        n = self.n_active
        if not self.hwc_direct:
            for b in range(n):
                ops.pack_features(feat[b:b + 1], out=self.feats_hwc[b])
        if self.tables is not None:  # f32 / f16x3 heads blend table rows (query_table.hip, query16.hip)
            # (re)made after every encoder pass; the handle of the previous pass unregisters the same
            # pointers only if they still point at its table views, so dropping it here is harmless
            self._table_handle = ops.skip_table_batch(mlp, self.feat_hwc_all[:n], out=self.tables[:n])
        # the octree of all frames of the slot level by level: one fused-query launch per level
        # covers every frame (mp_recon_batch takes up to MAX_RECON_BATCH frames per call)
        for b0 in range(0, n, MAX_RECON_BATCH):
            b1 = min(b0 + MAX_RECON_BATCH, n)
            ops.recon_batch(mlp, self.feats_hwc[b0:b1], self.calib[b0:b1], Z_SCALE, self.b_min,
                            self.b_max, self.res, self.balance, volumes=self.volumes[b0:b1],
                            status=self.status[b0:b1], final_level=self.final_level)
        pts_all = []
        # forward_vertices and the normal renders of all frames of the slot: one set of launches each
        # (mp_forward_vertices_batch, mp_paint_batch), results identical to the per-frame calls
        if VERTEX_BATCH:
            raws, renders = [], []
            for b0 in range(0, n, VERTEX_BATCH):
                rw = ops.forward_vertices_raw_batch(self.volumes[b0:min(b0 + VERTEX_BATCH, n)], "front")
                raws += rw
                renders += ops.paint_batch([v[0] for v in rw], [v[1] for v in rw], [v[3] for v in rw], 0,
                                           [v[4] for v in rw], r, 0.5, 0.5, 0.0, 1.0)
        else:  # A/B: one set of launches per frame
            raws = [ops.forward_vertices_raw(self.volumes[b], "front") for b in range(n)]
            renders = [ops.paint(v[0], v[1], v[3], 0, v[4], r, 0.5, 0.5, 0.0, 1.0) for v in raws]
        for b in range(n):
            x, y, z, nrm, count = raws[b]
            self.vertices[b] = raws[b]
            self.renders[b] = renders[b]
            if self.netC is not None:
                # netC.filter(image_c, feat_prior=featG_last) -> cat([prior, featC])
                # (MonoPortNet.py:41-45) packed straight into one channels-last map per frame
                ops.pack_features([feat[b:b + 1], feat_c[b:b + 1]], out=self.feats_hwc_c[b])
                pts_all.append(ops.vertex_points(x, y, z, count, r, self.mat_color))
        if self.netC is not None:
            # netC.query on the visible vertices (RTL/main.py:231-248) of ALL frames of the slot:
            # one fused-query launch per chunk of 32 frames (14 k points per frame alone would
            # leave most CUs idle)
            for b0 in range(0, n, MAX_RECON_BATCH):
                b1 = min(b0 + MAX_RECON_BATCH, n)
                preds = ops.query_counted_batch(
                    mlp_c, self.feats_hwc_c[b0:b1], pts_all[b0:b1],
                    [self.vertices[b][4] for b in range(b0, b1)], self.calib[b0:b1], Z_SCALE)
                tex = ops.paint_batch([self.vertices[b][0] for b in range(b0, b1)], [self.vertices[b][1] for b in range(b0, b1)],
                                      preds, 1, [self.vertices[b][4] for b in range(b0, b1)], r, 0.5, 0.5, -np.inf, np.inf)
                for b in range(b0, b1):
                    self.renders_tex[b] = tex[b - b0]

    def prepare(self, warmup=2):
        """Warm up (scratch arenas, GroupNorm accumulator arena); with ``use_graph`` capture the ENCODER into a
        hipGraph.  The C-ABI stages stay eager: they are a handful of asynchronous calls, and a
        graph that also holds them faulted on ROCm 7.2 once tensors were allocated after
        capture."""
        self.graph = None  # warm up eagerly with the CURRENT settings, then (re-)capture
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                self._chain()
        self.stream.synchronize()
        if self.use_graph:
            graph = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread (multi-GPU runs) and the other slots' host
            # threads may touch the HIP runtime while this thread captures
            with torch.cuda.graph(graph, stream=self.stream, capture_error_mode="thread_local"):
                self._graph_feat = self._encode()
            self.stream.synchronize()
            self.graph = graph
            # first launch of an instantiated graph uploads it: keep that out of the caller's first frame
            with torch.cuda.stream(self.stream):
                self.graph.replay()
            self.stream.synchronize()

    def submit(self, images, calibs, images_c=None):
        """Enqueue the reconstruction of n <= ``batch`` frames: ``images`` [n,3,512,512] (or a list
        of [1,3,512,512]), ``calibs`` [n,4,4] (or a list of [1,4,4]); returns immediately.  Results
        (``renders``, ``volumes``, ``status``, ``vertices``; entries 0..n-1) are valid after
        ``stream.synchronize()`` and until the next submit on this slot.  A short batch (the tail of
        a stream) still runs the encoder at the slot's batch size -- the graph is captured for it --
        on stale images in the unused entries; the octree and everything after it see n frames."""
        n = images.shape[0] if torch.is_tensor(images) else len(images)
        if not 1 <= n <= self.batch:
            raise ValueError("slot of %d frames got %d" % (self.batch, n))
        self.wait()  # a slot holds ONE batch: its previous results are overwritten from here on
        self.n_active = n
        # the caller may have produced the frames asynchronously on ITS stream (segmentation,
        # prepare_inputs, an H2D copy from pinned memory): order the slot's copies behind that work
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self._load(self.image, images)
            self._load(self.calib, calibs)
            if self.netC is not None:
                self._load(self.image_c, images if images_c is None else images_c)
            self._chain()
        self._busy = True

    def _load(self, dst, src):
        if torch.is_tensor(src):
            n = src.shape[0]
            dst[:n].copy_(src.reshape(dst[:n].shape), non_blocking=True)
            srcs = [src]
        else:
            srcs = list(src)
            if len(srcs) > 1 and all(s.is_cuda and s.device == dst.device and s.dtype == dst.dtype for s in srcs):
                # one gather launch for the frames of a submission instead of one copy per frame
                torch.cat([s.reshape(dst[:1].shape) for s in srcs], out=dst[:len(srcs)])
            else:
                for b, s in enumerate(srcs):
                    dst[b].copy_(s.reshape(dst[b].shape), non_blocking=True)
        for s in srcs:
            if s.is_cuda:  # the caching allocator must not recycle a source the copy still reads
                s.record_stream(self.stream)

    def wait(self):
        """Block the host until this slot's batch (and anything queued after it on the slot's
        stream before the next submit) has finished.  Results are to be consumed after ``wait()``
        or on a stream that has done ``wait_stream(slot.stream)``."""
        if self._busy:
            self.stream.synchronize()
            self._busy = False

    def close(self):
        """Release the C-ABI scratch arena keyed by this slot's stream (mp_stream_release)."""
        self.wait()
        self.graph = None
        if self.tables is not None:
            if self._table_handle is not None:
                self._table_handle.release()
            self._table_handle = None
            self.tables = None
        ops.stream_release(self.stream)


class FramePipeline:
    """Round-robin over ``depth`` FrameSlots of ``batch`` frames each."""

    def __init__(self, netG, device, depth=2, batch=1, **slot_kwargs):
        self.slots = [FrameSlot(netG, device, batch=batch, **slot_kwargs) for _ in range(depth)]
        self.batch = int(batch)
        self.n_submitted = 0

    def prepare(self):
        for s in self.slots:
            s.prepare()

    def submit(self, images, calibs, images_c=None):
        slot = self.slots[self.n_submitted % len(self.slots)]
        slot.submit(images, calibs, images_c)
        self.n_submitted += 1
        return slot

    def synchronize(self):
        for s in self.slots:
            s.wait()
            s.stream.synchronize()

    def close(self):
        for s in self.slots:
            s.close()
