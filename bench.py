#!/usr/bin/env python
"""Headline benchmark: reconstructions/sec (512x512 image in, 256^3-effective octree grid out).

One "step" = one full geometry reconstruction of one synthetic frame on one MI355X
(BASELINE.json configs[1]): netG.filter (hourglass encoder, PyTorch-ROCm) -> channels-last pack
-> coarse-to-fine octree 17..257 driving the fused HIP query kernel -> forward_vertices ->
normal render.  Inputs are resident in HBM before the timed region.  With --gpus N every rank
reconstructs its own frames (frame-parallel, weak scaling) and the renders are gathered to rank 0
over RCCL.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     the fused query kernel against the f32 MFMA peak (HIP-event timed, live)
  "cpu_baseline": the CPU oracle path timed on this box's host cores (rank 0, N=1 only)

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from monoport_amd import ops, parallel, synthetic as syn  # noqa: E402
from monoport_amd.modeling import PIFuNetC, PIFuNetG  # noqa: E402
from monoport_amd.pipeline import FramePipeline  # noqa: E402
from monoport_amd.recon import pifu_calib  # noqa: E402

RESOLUTIONS = [17, 33, 65, 129, 257]  # RTL/main.py:187
B_MIN, B_MAX = [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]  # RTL/main.py:185-186
FLOP_PER_POINT = 2363906  # netG MLP, SURVEY.md section 8d / BASELINE.md section 2
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def set_precision_everywhere(head, precision):
    """MLP arithmetic of the fused query kernel AND of the encoder's fused 3x3 convolutions:
    "f16x3" switches both to f32 emulated on f16 MFMA (three MFMAs per product, f32 accumulate);
    the other f16 query variants leave the encoder on exact f32."""
    from monoport_amd.modeling import backbones
    head.set_precision(precision)
    backbones.set_encoder_conv_precision("f16x3" if precision == "f16x3" else "f32")


def build_netg(device, precision="f32"):
    """Random-init (seeded) encoder of the reference architecture + the analytic F-body head."""
    net = PIFuNetG().eval()
    set_precision_everywhere(net.surface_classifier, precision)
    shapes = {k: tuple(v.shape) for k, v in net.image_filter.state_dict().items()}
    sd = syn.seeded_state_dict(shapes, 71)
    net.image_filter.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    layers = syn.body_mlp("G", noise=0.05, seed=1)
    net.surface_classifier.load_state_dict(
        {**{"filters.%d.weight" % i: torch.from_numpy(w)[:, :, None] for i, (w, _) in enumerate(layers)},
         **{"filters.%d.bias" % i: torch.from_numpy(b) for i, (_, b) in enumerate(layers)}})
    return net.to(device), layers


def build_netc(device):
    """netC with seeded random weights of the reference architecture (config 3)."""
    net = PIFuNetC().eval()
    shapes = {k: tuple(v.shape) for k, v in net.image_filter.state_dict().items()}
    sd = syn.seeded_state_dict(shapes, 72)
    net.image_filter.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    layers = syn.rand_mlp("C", 61, 2.0)
    net.surface_classifier.load_state_dict(
        {**{"filters.%d.weight" % i: torch.from_numpy(w)[:, :, None] for i, (w, _) in enumerate(layers)},
         **{"filters.%d.bias" % i: torch.from_numpy(b) for i, (_, b) in enumerate(layers)}})
    return net.to(device)


def make_pipeline(device, depth, use_graph, resolutions=None, with_color=False, precision="f32",
                  batch=1):
    """`depth` slots of `batch` frames each (monoport_amd/pipeline.py): per slot the batched
    encoder (a hipGraph unless --no-graph), then the stage chain of RTL/main.py:389-428 as
    asynchronous C-ABI calls on the slot's stream."""
    net, _ = build_netg(device, precision)
    planes = torch.from_numpy(syn.body_feature_planes(128, 128)).to(device)

    def body_planes_hook(feat):
        # synthetic-data hook: the analytic F-body head reads channels 0/1 as depth planes; the
        # other 254 channels are the encoder's output (consumed through the seeded-noise weights)
        feat[:, 0:2].copy_(planes[None].expand(feat.shape[0], -1, -1, -1))

    planes_hwc = planes.permute(1, 2, 0).contiguous()

    def body_planes_hook_hwc(feat_hwc):  # the same edit on the channels-last [B,H,W,C] map
        feat_hwc[..., 0:2].copy_(planes_hwc[None].expand(feat_hwc.shape[0], -1, -1, -1))

    body_planes_hook.hwc = body_planes_hook_hwc

    pipe = FramePipeline(net, device, depth=depth, batch=batch, resolutions=resolutions or RESOLUTIONS,
                         b_min=B_MIN, b_max=B_MAX, balance=0.5, feature_hook=body_planes_hook,
                         use_graph=use_graph, netC=build_netc(device) if with_color else None)
    pipe.prepare()
    return pipe


TRAFFIC_PROFILE = "r02_query_traffic.json"


def traffic_from_profile(args, frames_per_launch):
    """HBM-side bytes per fused-query launch from the committed PMC pass (separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE runs of tools/traffic_probe.py, corrected as
    MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside this process, so the
    figure is reported ONLY for the configuration the pass covered (f32 kernel, 5 levels, geometry
    only, same frames per launch) and is None for every other run or when the profile is absent."""
    if args.precision != "f32" or args.levels != 5 or args.with_color:
        return None
    path = os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)
    try:
        with open(path) as f:
            prof = json.load(f)
        if int(prof["frames_per_launch"]) != int(frames_per_launch):
            return None
        return prof["bytes_per_launch_avg"]
    except (OSError, KeyError, ValueError):
        return None


def dropin_surface(device, n_frames, n_warm, resolutions):
    """The reference's own call surface, as RTL/main.py:326-452 drives it: the processors=[...]
    list (H2D, camera, pifu_calib, input normalisation, netG.filter, reconEngine =
    Seg3dLossless(query_func) with its per-frame host sync, forward_vertices with its .item(),
    colorization) on the thread-per-stage pipeline -- one frame per call, batch 1, eager encoder.
    Returns recon/s through that surface and the latency of a single frame run stage by stage."""
    from monoport_amd.implicit_seg.functional import Seg3dLossless
    from monoport_amd.recon import colorization, forward_vertices
    from monoport_amd.stage_pipeline import StagePipeline
    netG, _ = build_netg(device)
    planes = torch.from_numpy(syn.body_feature_planes(128, 128)).to(device)

    def query_func(points, im_feat_list, calib_tensor):  # RTL/main.py:169-183
        assert len(points) == 1
        samples = points.repeat(1, 1, 1)
        samples = samples.permute(0, 2, 1)
        return netG.query(im_feat_list, points=samples, calibs=calib_tensor)[0]

    engine = Seg3dLossless(query_func=query_func, b_min=np.array([B_MIN], np.float32),
                           b_max=np.array([B_MAX], np.float32), resolutions=resolutions,
                           balance_value=0.5, use_cuda_impl=False, faster=True).to(device)
    mean, std = 0.5, 0.5
    r_last = resolutions[-1]

    def filt(d):
        feats = netG.filter(d["input_netG"])
        feats[-1][0][0, 0:2].copy_(planes)  # synthetic body planes, as in the headline run
        return {**d, "feat_tensor_G": feats}

    def processors(step):
        def camera(d):
            ext, intr = syn.scene_camera(3 * step[0])
            step[0] += 1
            return {**d, "extrinsic": ext, "intrinsic": intr}
        return [
            lambda data: {"input": data.to(device, non_blocking=True)},                    # main.py:327
            camera,                                                                       # :330-336
            lambda d: {**d, "calib_tensor": pifu_calib(d["extrinsic"], d["intrinsic"], device=device)},
            lambda d: {**d, "input_netG": (((d["input"][:, 0:3] * 0.5 + 0.5) - mean) / std)
                       * d["input"][:, 3:4]},                                             # :353-357
            filt,                                                                         # :367-370
            lambda d: {**d, "sdf": engine(im_feat_list=d["feat_tensor_G"],
                                          calib_tensor=d["calib_tensor"])},               # :390-395
            lambda d: {**d, **dict(zip(["X", "Y", "Z", "norm"],
                                       forward_vertices(d["sdf"], direction="front")))},  # :401-406
            lambda d: {**d, "render_norm": colorization(None, None, d["X"], d["Y"], d["Z"],
                                                        d["calib_tensor"], d["norm"],
                                                        resolution=r_last)},              # :418-428
        ]

    frames = []
    for i in range(4):
        img = torch.from_numpy(syn.synthetic_image(i))
        mask = (img.abs().sum(0, keepdim=True) > 0).float()
        frames.append(torch.cat([img, mask], 0)[None].pin_memory())

    # single-frame latency: one frame through the stages, one after the other, nothing else on
    # the GPU; median of 5 after a warm-up
    procs = processors([0])
    lat = []
    with torch.no_grad():
        for i in range(2 + 5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d = frames[i % 4]
            for p in procs:
                d = p(d)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
    assert d["render_norm"] is not None
    latency_ms = float(np.median(lat[2:])) * 1e3

    # throughput: the same list on the stage pipeline (thread + stream per stage, FIFO order)
    def source():
        for i in range(n_warm + n_frames):
            yield frames[i % 4]

    out_count, t0 = 0, None
    with torch.no_grad():
        for d in StagePipeline(source(), processors([0]), device=device, max_in_flight=8):
            out_count += 1
            if out_count == n_warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert out_count == n_warm + n_frames and engine.last_path == "fused"
    return {
        "surface": "RTL/main.py processors list on StagePipeline: Seg3dLossless(query_func) + "
                   "forward_vertices + colorization, batch 1, eager encoder, 8 frames in flight",
        "value": n_frames / elapsed, "unit": "recon/s", "ms_per_step": elapsed / n_frames * 1e3,
        "latency_ms_single_frame": latency_ms,
        "frames": n_frames,
    }


def rank_devices(dist, device, world):
    """(device index, PCI bus id, uuid-ish name) of every rank, gathered on all ranks."""
    props = torch.cuda.get_device_properties(device)
    bus = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1),
                              getattr(props, "pci_device_id", -1))
    mine = "%d|%s|%s" % (torch.cuda.current_device(), bus, props.name)
    if dist is None:
        return [mine]
    got = [None] * world
    dist.all_gather_object(got, mine)
    return got


def cpu_baseline(threads):
    """One reconstruction on the host cores: encoder (torch CPU, the reference's own op set) +
    CPU oracle octree / query / forward_vertices.  Test infrastructure used as the baseline."""
    from oracle import pifu_oracle as orc
    torch.set_num_threads(threads)
    net, layers = build_netg("cpu")
    img = torch.from_numpy(syn.synthetic_image(0))[None]
    calib = orc.pifu_calib(*syn.scene_camera(0))[0]
    planes = syn.body_feature_planes(128, 128)
    t0 = time.perf_counter()
    with torch.no_grad():
        feat = net.image_filter(img, last_only=True)[-1][0][0].numpy().copy()
    feat[0:2] = planes
    t1 = time.perf_counter()
    stats = []
    vol = orc.seg3d_lossless(
        lambda p: orc.query(feat, p, calib, layers, 1, syn.Z_SCALE, precision="f32", threads=threads)[0],
        B_MIN, B_MAX, RESOLUTIONS, stats=stats)
    t2 = time.perf_counter()
    orc.forward_vertices(vol, "front")
    t3 = time.perf_counter()
    total = t3 - t0
    return {
        "value": 1.0 / total, "unit": "recon/s", "cores": threads, "kind": "port",
        "sample": "1 reconstruction: encoder %.2fs (torch CPU) + octree %.2fs (%d pts, C oracle f32, "
                  "OpenMP) + forward_vertices %.2fs" % (t1 - t0, t2 - t1, sum(stats), t3 - t2),
        "mpts_per_s": sum(stats) / (t2 - t1) / 1e6,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--depth", type=int, default=3, help="pipeline slots (streams) per GPU")
    ap.add_argument("--batch", type=int, default=10,
                    help="frames per slot (upper bound): their encoder passes run as one batch and "
                         "their octree levels as one fused-query launch; depth x batch frames are in "
                         "flight.  The largest divisor of --steps not above this is used, so no slot "
                         "submission is short (a short batch would still pay the full-batch encoder)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch the encoder eagerly instead of replaying it as a hipGraph")
    ap.add_argument("--with-color", action="store_true",
                    help="BASELINE configs[2]: add netC (ResNet encoder + per-vertex colour MLP)")
    ap.add_argument("--levels", type=int, default=5, choices=[5, 6],
                    help="6 = octree to 513^3 (BASELINE configs[4] grid, f32 weights)")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3", "f16w", "f16"],
                    help="MLP arithmetic: exact f32 MFMA (default); f16x3 = f32-accurate 3-term f16 "
                         "split (hi*hi + hi*lo + lo*hi on f16 MFMA, f32 accumulate); f16w = fp16 "
                         "weights, split activations (BASELINE configs[4]); f16 = fp16 operands")
    ap.add_argument("--mode", default="pipeline", choices=["pipeline", "dropin"],
                    help="pipeline (default): the headline -- FramePipeline, frames batched per slot, no "
                         "host sync; dropin: `value` is measured through the reference's call surface "
                         "(StagePipeline + Seg3dLossless + forward_vertices), as the default run's "
                         "`dropin` object")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the drop-in-surface pass a default N=1 run appends")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the informational f16x3 pass that a default N=1 run appends")
    args = ap.parse_args()

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/test_dropin_gpu.py): exercise the N > 1 code path on a ONE-GPU box -- every
    # rank on device 0, collectives over gloo instead of RCCL (which refuses two ranks per device)
    one_gpu_test = os.environ.get("MONOPORT_BENCH_ONE_GPU_TEST") == "1"
    if one_gpu_test:
        local_rank = 0
    if int(os.environ.get("WORLD_SIZE", "1")) not in (1, args.gpus):
        raise SystemExit("--gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ["WORLD_SIZE"]))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rank, world = parallel.init_from_env(backend="gloo" if one_gpu_test else "nccl",
                                         device=device)  # nccl = RCCL on ROCm
    dist = None
    if world > 1:
        import torch.distributed as dist
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    # every rank on its own GPU: (device index, PCI bus id) must be N distinct values (the one-GPU
    # test hook deliberately shares device 0)
    devices = rank_devices(dist, device, world)
    if world > 1 and not one_gpu_test:
        assert len(set(devices)) == world, "ranks share a GPU: %s" % devices

    resolutions = RESOLUTIONS + ([513] if args.levels == 6 else [])
    if args.mode == "dropin":
        assert world == 1, "--mode dropin is a single-GPU measurement"
        res = dropin_surface(device, args.steps, args.warmup, resolutions)
        print(json.dumps({
            "metric": "reconstructions/sec (512^2 in, %d^3 grid) through the drop-in surface" % (resolutions[-1] - 1),
            "value": res["value"], "unit": "recon/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": res["surface"]},
            "latency_ms_single_frame": res["latency_ms_single_frame"]}), flush=True)
        return
    batch = max(b for b in range(1, max(1, min(args.batch, args.steps)) + 1) if args.steps % b == 0)
    use_graph = not args.no_graph
    try:
        pipe = make_pipeline(device, args.depth, use_graph, resolutions, args.with_color,
                             args.precision, batch)
    except RuntimeError as e:
        # hipGraph capture can fail next to an initialised RCCL communicator (its watchdog thread
        # touches the runtime): fall back to eager encoder launches rather than lose the run
        if not use_graph:
            raise
        sys.stderr.write("bench: hipGraph capture failed (%s); falling back to --no-graph\n" % e)
        torch.cuda.synchronize()
        use_graph = False
        pipe = make_pipeline(device, args.depth, False, resolutions, args.with_color,
                             args.precision, batch)
    if dist is not None:  # all ranks run the same variant
        flag = torch.tensor([int(use_graph)], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if use_graph and int(flag.item()) == 0:
            use_graph = False
            pipe = make_pipeline(device, args.depth, False, resolutions, args.with_color,
                                 args.precision, batch)
    n_warm = args.warmup
    n_frames = args.steps + n_warm
    # distinct frames per rank: frame id = step * world + rank (frame-parallel sharding)
    images = [torch.from_numpy(syn.synthetic_image(s * world + rank))[None].to(device)
              for s in range(min(n_frames, 4))]
    calibs = [pifu_calib(*syn.scene_camera(3 * (s * world + rank)), device=device)
              for s in range(n_frames)]
    r_last = resolutions[-1]
    # one gather per slot submission: [batch, R, R, 3] renders to rank 0 (a no-op on one GPU)
    gather = parallel.FrameGather((batch, r_last, r_last, 3), device=device, store=False)
    gather_checked = [False]
    render_pack = [torch.zeros((batch, r_last, r_last, 3), dtype=torch.float32, device=device)
                   for _ in range(args.depth)]
    status_log = []

    def run_batch(pipe, s0, s1, log):
        """Frames s0 .. s1-1 (at most `batch`) as one slot submission."""
        slot = pipe.submit([images[s % len(images)] for s in range(s0, s1)],
                           [calibs[s] for s in range(s0, s1)])
        with torch.cuda.stream(slot.stream):
            if world > 1:
                pack = render_pack[(pipe.n_submitted - 1) % args.depth]  # this slot's staging buffer
                for b in range(s1 - s0):
                    pack[b].copy_(slot.renders_tex[b] if args.with_color else slot.renders[b])
                gather.push(s0 // batch, pack)
                if not log and rank == 0 and not gather_checked[0]:
                    # (warm-up only: this syncs) the gathered copy of rank 0's own frames must
                    # equal what rank 0 rendered
                    got = gather.received(0)[:s1 - s0].to(pack.device)
                    assert torch.equal(got, pack[:s1 - s0]), "gather mismatch"
                    gather_checked[0] = True
            if log:
                status_log.append(slot.status[:s1 - s0].clone())  # device-side copy, no sync

    def timed_pass(pipe, log):
        """Warm-up batches, then EXACTLY --steps frames between barrier + synchronize pairs."""
        for s0 in range(0, n_warm, batch):
            run_batch(pipe, s0, min(s0 + batch, n_warm), False)
        pipe.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for s0 in range(n_warm, n_frames, batch):
            run_batch(pipe, s0, min(s0 + batch, n_frames), log)
        pipe.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0

    elapsed = timed_pass(pipe, True)

    # informational second pass (N=1 only, never `value`): the same frames with the MLP on the
    # f32-accurate f16x3 kernel, plus the largest difference between the two volumes of one frame
    alt = None
    if world == 1 and args.precision == "f32" and not args.no_alt and not args.with_color:
        last_slot = pipe.slots[(pipe.n_submitted - 1) % len(pipe.slots)]
        vol_f32 = last_slot.volumes[last_slot.n_active - 1].clone()
        # a SECOND pipeline (own network copy, own captured graphs): the f32 pipeline and its graphs
        # stay untouched for the roofline / breakdown legs below
        pipe16 = make_pipeline(device, args.depth, use_graph, resolutions, False, "f16x3", batch)
        alt_elapsed = timed_pass(pipe16, False)
        last_slot = pipe16.slots[(pipe16.n_submitted - 1) % len(pipe16.slots)]
        vol_alt = last_slot.volumes[last_slot.n_active - 1]
        diff = (vol_alt - vol_f32).abs().max().item()
        flips = int(((vol_alt > 0.5) != (vol_f32 > 0.5)).sum().item())
        pipe16.close()
        del pipe16, last_slot, vol_alt
        from monoport_amd.modeling import backbones
        backbones.set_encoder_conv_precision("f32")  # process-wide switch back for the eager legs
        torch.cuda.synchronize()
        alt = {"precision": "f16x3 (f32 emulated on f16 MFMA, 3-term split, f32 accumulate) in the query "
                            "kernel AND in the encoder's 3x3 convolutions",
               "value": args.steps / alt_elapsed, "unit": "recon/s",
               "ms_per_step": alt_elapsed / args.steps * 1e3,
               "max_abs_diff_vs_f32_volume": diff, "thresholded_voxels_differing": flips,
               "voxels": int(vol_f32.numel()),
               "note": "opt-in (--precision f16x3); not the headline"}

    # roofline leg: the same frames again on ONE stream with every fused-query launch bracketed by
    # HIP events on its launch stream (concurrent slots would share CUs) -> per-launch durations
    # of the dominant kernel
    from monoport_amd.pipeline import MAX_RECON_BATCH
    prof_slot = pipe.slots[0]
    prof_status = []
    ops.profile_begin(device, max_records=8 * args.steps + 8)
    for s0 in range(n_warm, n_frames, batch):
        s1 = min(s0 + batch, n_frames)
        prof_slot.submit([images[s % len(images)] for s in range(s0, s1)],
                         [calibs[s] for s in range(s0, s1)])
        with torch.cuda.stream(prof_slot.stream):
            prof_status.append(prof_slot.status[:s1 - s0].clone())
    prof_slot.wait()
    all_ms = ops.profile_end(device, capacity=8 * args.steps + 8)
    # launch order per submission of n frames: for every chunk of <= 8 frames one launch per level
    # (its points = that level's nodes summed over the chunk, pipeline.py / mp_recon_batch); with
    # --with-color one netC launch per chunk follows, which the netG roofline skips
    launch_ms, prof_pts, cursor = [], [], 0
    for st in prof_status:
        counts = st.cpu().numpy()[:, 1:]
        for b0 in range(0, counts.shape[0], MAX_RECON_BATCH):
            prof_pts.append(counts[b0:b0 + MAX_RECON_BATCH].sum(0))
            launch_ms.append(all_ms[cursor:cursor + len(resolutions)])
            cursor += len(resolutions)
        if args.with_color:  # one netC launch per chunk of <= 8 frames follows (skipped here)
            cursor += (counts.shape[0] + MAX_RECON_BATCH - 1) // MAX_RECON_BATCH
    launch_ms = np.concatenate(launch_ms)
    prof_pts = np.stack(prof_pts)

    # breakdown leg: encoder-only and encoder-excluded time per frame, one
    # stream, features of the last frame
    def timed(fn, n):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(prof_slot.stream):
            fn()
            ev0.record(prof_slot.stream)
            for _ in range(n):
                fn()
            ev1.record(prof_slot.stream)
        prof_slot.stream.synchronize()
        return ev0.elapsed_time(ev1) / n

    def recon_only():
        mlp = prof_slot.net.surface_classifier.packed()
        ops.recon(mlp, prof_slot.feats_hwc[0], prof_slot.calib[0:1], syn.Z_SCALE, B_MIN, B_MAX,
                  resolutions, 0.5, volume=prof_slot.volume, status=prof_slot.status[0])
        x, y, z, nrm, count = ops.forward_vertices_raw(prof_slot.volume, "front")
        ops.paint(x, y, nrm, 0, count, r_last, 0.5, 0.5, 0.0, 1.0)

    def recon_batched():
        """What a slot does after its encoder: the octree of its frames level by level, then per
        frame forward_vertices + render (monoport_amd/pipeline.py)."""
        mlp = prof_slot.net.surface_classifier.packed()
        nb = min(batch, MAX_RECON_BATCH)
        ops.recon_batch(mlp, prof_slot.feats_hwc[:nb], prof_slot.calib[:nb], syn.Z_SCALE, B_MIN, B_MAX,
                        resolutions, 0.5, volumes=prof_slot.volumes[:nb], status=prof_slot.status[:nb])
        for b in range(nb):
            x, y, z, nrm, count = ops.forward_vertices_raw(prof_slot.volumes[b], "front")
            ops.paint(x, y, nrm, 0, count, r_last, 0.5, 0.5, 0.0, 1.0)

    with torch.no_grad():
        # the last submission may have been a short batch: refill the slot so every entry is live
        prof_slot.submit([images[s % len(images)] for s in range(batch)], calibs[:batch])
        prof_slot.wait()
        enc_ms = timed(lambda: prof_slot.net.image_filter(prof_slot.image, last_only=True), 10) / batch
        rec_ms = timed(recon_only, 10)
        rec_batched_ms = timed(recon_batched, 5) / min(batch, MAX_RECON_BATCH)

    statuses = torch.cat(status_log).cpu().numpy()
    assert (statuses[:, 0] == 1).all(), "synthetic body must be non-empty"
    level_pts = statuses[:, 1:]
    pts_total = int(level_pts.sum())
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    p = torch.tensor([pts_total], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    pts_all = float(p.item())

    if rank == 0:
        # the f16 variants spend 3 / 2 / 1 f16 MFMAs (2.5 PFLOP/s dense peak) per algorithmic product
        terms = {"f32": 0, "f16x3": 3, "f16w": 2, "f16": 1}[args.precision]
        peak_tflops = F32_MFMA_PEAK_TFLOPS if terms == 0 else 2500.0 / terms
        n_launch = min(len(launch_ms), prof_pts.size)
        flops = prof_pts.reshape(-1)[:n_launch].astype(np.float64) * FLOP_PER_POINT
        achieved = flops.sum() / (launch_ms[:n_launch].sum() * 1e-3) / 1e12 if n_launch else 0.0
        out = {
            "metric": "reconstructions/sec (512^2 in, %d^3 grid)" % (r_last - 1),
            "value": args.steps * world / elapsed,
            "unit": "recon/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32",
                      "f16x3": "f32 emulated on f16 MFMA (3-term hi/lo split, f32 accumulate)",
                      "f16w": "f16 weights x split-f16 activations, f32 accumulate",
                      "f16": "f16 operands, f32 accumulate"}[args.precision],
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[%d]: single 512x512 image, netG (4-stack hourglass encoder "
                             "fp32 + fused query), octree %s on [-1,1]^3, %s"
                             % (2 if args.with_color else (4 if args.levels == 6 else 1),
                                "-".join(str(r) for r in resolutions),
                                "geometry + netC per-vertex colour (ResNet encoder + colour MLP)"
                                if args.with_color else
                                "geometry only (+forward_vertices, normal render)")),
                "frames_per_rank": args.steps,
                "devices": devices,
                "gather_checked": bool(gather_checked[0]) if world > 1 else None,
                "parallelism": "frame-parallel x%d (one process per GPU, renders gathered to rank 0 "
                               "over %s); per GPU %d slots x %d frames in flight, encoder "
                               "batched per slot%s"
                               % (world, "gloo (one-GPU test hook)" if one_gpu_test else "RCCL",
                                  args.depth, batch,
                                  " and replayed as a hipGraph" if use_graph else ""),
                "fixture": "F-body analytic head, seeded encoder (monoport_amd/synthetic.py)",
                "points_per_recon": pts_all / (args.steps * world),
            },
            "mpts_per_s": pts_all / elapsed / 1e6,
            "breakdown": {
                "encoder_ms_per_frame": enc_ms, "recon_vertices_render_ms": rec_ms,
                "recon_vertices_render_ms_per_frame_batched": rec_batched_ms,
                "recon_per_s_encoder_excluded": 1e3 / rec_batched_ms,
                "recon_per_s_encoder_excluded_single_frame": 1e3 / rec_ms,
                "points_per_level": [float(v) / args.steps for v in prof_pts.sum(0)],
                "note": "single stream, no overlap; encoder eager at the bench batch size; batched = "
                        "mp_recon_batch over the slot's frames, as the pipeline runs it",
            },
            "roofline": {
                "kernel": ("pifu_query_kernel<256,1> (fused gather + MLP)" if args.precision == "f32"
                           else "pifu_query16_kernel<1,%d> (fused gather + MLP, %s)" % (terms, args.precision)),
                "bound": "mfma",
                "achieved": achieved,
                "peak": peak_tflops,
                "unit": "TFLOP/s",
                "frac": achieved / peak_tflops,
                "traffic": traffic_from_profile(args, min(batch, MAX_RECON_BATCH)),
                "traffic_source": ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                   "this configuration)" % TRAFFIC_PROFILE),
                "launches": int(n_launch),
                "frames_per_launch": min(batch, MAX_RECON_BATCH),
                "avg_launch_ms": float(launch_ms[:n_launch].mean()) if n_launch else None,
                "flop_per_point": FLOP_PER_POINT,
            },
        }
        if alt is not None:
            out["alt_precision"] = alt
        if world == 1 and not args.no_dropin and not args.with_color and args.precision == "f32":
            out["dropin"] = dropin_surface(device, args.steps, args.warmup, resolutions)
            out["latency_ms_single_frame"] = out["dropin"]["latency_ms_single_frame"]
        if world == 1 and not args.no_cpu_baseline and not args.with_color and args.levels == 5:
            # bounded thread count: torch-CPU convs at batch 1 collapse when oversubscribed
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 32))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
