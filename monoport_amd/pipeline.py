"""Per-frame reconstruction pipeline with several frames in flight on one GPU.

The reference overlaps frames with one Python thread per stage (RTL/dataloader.py:734-751,
:1026-1053).  On an MI355X the stages are so short (tens of microseconds to a few milliseconds)
that host-side launch latency, not thread parallelism, is what matters: here each in-flight frame
owns a HIP stream and a hipGraph holding its whole stage chain

    netG.filter -> channels-last pack -> octree (5 levels, fused query) -> forward_vertices -> render

(optional: `use_graph=True`; eager launches from one host thread are within ~4 % because every
C-ABI call is asynchronous), and independent frames on different streams fill the CUs that the
coarse octree levels and the small encoder kernels leave idle.
"""
import numpy as np
import torch

from . import ops
from .synthetic import Z_SCALE

RESOLUTIONS = (17, 33, 65, 129, 257)  # RTL/main.py:187


class FrameSlot:
    """Static buffers + captured graph for one in-flight frame (geometry-only chain,
    RTL/main.py:366-428)."""

    def __init__(self, netG, device, resolutions=RESOLUTIONS, b_min=(-1, -1, -1), b_max=(1, 1, 1),
                 balance=0.5, feature_hook=None, use_graph=False, netC=None):
        self.net = netG
        self.netC = netC  # optional colour network: adds the texture stages of RTL/main.py:373-441
        self.device = torch.device(device)
        self.res = [int(r) for r in resolutions]
        self.b_min, self.b_max, self.balance = b_min, b_max, float(balance)
        self.feature_hook = feature_hook  # optional in-place edit of the [1,C,H,W] feature map
        r = self.res[-1]
        dev = self.device
        self.stream = torch.cuda.Stream(device=dev)
        self.image = torch.zeros((1, 3, 512, 512), dtype=torch.float32, device=dev)
        self.calib = torch.eye(4, dtype=torch.float32, device=dev)[None].contiguous()
        self.feat_hwc = torch.empty((128, 128, 256), dtype=torch.float32, device=dev)
        self.volume = torch.empty((r, r, r), dtype=torch.float32, device=dev)
        self.status = torch.zeros((1 + len(self.res),), dtype=torch.int32, device=dev)
        self.render = None
        self.render_tex = None
        self.vertices = None
        if netC is not None:
            from .recon import color_matrix
            self.feat_hwc_c = torch.empty((128, 128, 512), dtype=torch.float32, device=dev)
            self.mat_color = color_matrix(b_min, b_max, r)
            self.image_c = torch.zeros((1, 3, 512, 512), dtype=torch.float32, device=dev)
        self.graph = None
        self.use_graph = use_graph
        self.done = torch.cuda.Event()  # recorded after each frame's last kernel
        self._busy = False

    @torch.no_grad()
    def _encode(self):
        feat = self.net.image_filter(self.image, last_only=True)[-1][0]
        if self.feature_hook is not None:
            self.feature_hook(feat)
        return feat

    @torch.no_grad()
    def _chain(self):
        mlp = self.net.surface_classifier.packed()
        if self.graph is not None:
            self.graph.replay()  # the ~450 encoder kernels as one hipGraph launch
            feat = self._graph_feat
        else:
            feat = self._encode()
        ops.pack_features(feat, out=self.feat_hwc)
        ops.recon(mlp, self.feat_hwc, self.calib, Z_SCALE, self.b_min, self.b_max, self.res,
                  self.balance, volume=self.volume, status=self.status)
        x, y, z, nrm, count = ops.forward_vertices_raw(self.volume, "front")
        self.vertices = (x, y, z, nrm, count)
        self.render = ops.paint(x, y, nrm, 0, count, self.res[-1], 0.5, 0.5, 0.0, 1.0)
        if self.netC is not None:
            # netC.filter(image_c, feat_prior=featG_last) -> cat([prior, featC]) (MonoPortNet.py:41-45)
            # packed straight into one channels-last map, then netC.query on the visible vertices
            mlp_c = self.netC.surface_classifier.packed()
            feat_c = self.netC.image_filter(self.image_c)[-1][0]
            ops.pack_features([feat, feat_c], out=self.feat_hwc_c)
            pts = ops.vertex_points(x, y, z, count, self.res[-1], self.mat_color)
            preds = ops.query_counted(mlp_c, self.feat_hwc_c, pts, count, self.calib, Z_SCALE)
            self.render_tex = ops.paint(x, y, preds, 1, count, self.res[-1], 0.5, 0.5,
                                        -np.inf, np.inf)

    def prepare(self, warmup=2):
        """Warm up (MIOpen find, scratch arenas); with ``use_graph`` capture the ENCODER into a
        hipGraph (its ~450 small kernels are launch-latency bound: 5.7 -> 4.8 ms).  The C-ABI
        stages stay eager: they are five asynchronous calls, and a graph that also holds them
        faulted on ROCm 7.2 once tensors were allocated after capture."""
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                self._chain()
        self.stream.synchronize()
        if self.use_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=self.stream):
                self._graph_feat = self._encode()
            self.stream.synchronize()
            self.graph = graph

    def submit(self, image, calib, image_c=None):
        """Enqueue one reconstruction of ``image`` [1,3,512,512] with ``calib`` [1,4,4]; returns
        immediately.  Results (``render``, ``volume``, ``status``, ``vertices``) are valid after
        ``stream.synchronize()`` and until the next submit on this slot."""
        self.wait()  # a slot holds ONE frame: its previous results are overwritten from here on
        with torch.cuda.stream(self.stream):
            self.image.copy_(image, non_blocking=True)
            self.calib.copy_(calib, non_blocking=True)
            if self.netC is not None:
                self.image_c.copy_(image if image_c is None else image_c, non_blocking=True)
            self._chain()
            self.done.record(self.stream)
        self._busy = True

    def wait(self):
        """Block the host until this slot's frame (and anything queued after it on the slot's
        stream before the next submit) has finished."""
        if self._busy:
            self.stream.synchronize()
            self._busy = False


class FramePipeline:
    """Round-robin over ``depth`` FrameSlots: frame i runs on slot i % depth."""

    def __init__(self, netG, device, depth=2, **slot_kwargs):
        self.slots = [FrameSlot(netG, device, **slot_kwargs) for _ in range(depth)]
        self.n_submitted = 0

    def prepare(self):
        for s in self.slots:
            s.prepare()

    def submit(self, image, calib, image_c=None):
        slot = self.slots[self.n_submitted % len(self.slots)]
        slot.submit(image, calib, image_c)
        self.n_submitted += 1
        return slot

    def synchronize(self):
        for s in self.slots:
            s.stream.synchronize()
