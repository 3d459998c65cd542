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


def build_netg(device, precision="f32"):
    """Random-init (seeded) encoder of the reference architecture + the analytic F-body head."""
    net = PIFuNetG().eval()
    net.surface_classifier.set_precision(precision)
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

    pipe = FramePipeline(net, device, depth=depth, batch=batch, resolutions=resolutions or RESOLUTIONS,
                         b_min=B_MIN, b_max=B_MAX, balance=0.5, feature_hook=body_planes_hook,
                         use_graph=use_graph, netC=build_netc(device) if with_color else None)
    pipe.prepare()
    return pipe


def traffic_from_profile(frames_per_launch):
    """HBM-side bytes per fused-query launch from the committed PMC pass (separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE runs of tools/traffic_probe.py, corrected as
    MI355X_MICROARCH.md prescribes); None if the profile is absent."""
    path = os.path.join(ROOT, "profiles", "r01e_query_traffic.json")
    try:
        with open(path) as f:
            prof = json.load(f)
        # the profile was taken at 4 frames per launch; traffic scales with the points of a launch
        return prof["bytes_per_launch_avg"] * frames_per_launch / prof["frames_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


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
    ap.add_argument("--batch", type=int, default=4,
                    help="frames per slot: their encoder passes run as one batch; depth x batch "
                         "frames are in flight (reduced to a divisor of --steps)")
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

    resolutions = RESOLUTIONS + ([513] if args.levels == 6 else [])
    batch = max(1, min(args.batch, args.steps))  # a pass's last batch may be shorter
    pipe = make_pipeline(device, args.depth, not args.no_graph, resolutions, args.with_color,
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
    render_pack = [torch.zeros((batch, r_last, r_last, 3), dtype=torch.float32, device=device)
                   for _ in range(args.depth)]
    status_log = []

    def run_batch(s0, s1, log):
        """Frames s0 .. s1-1 (at most `batch`) as one slot submission."""
        slot = pipe.submit([images[s % len(images)] for s in range(s0, s1)],
                           [calibs[s] for s in range(s0, s1)])
        with torch.cuda.stream(slot.stream):
            if world > 1:
                pack = render_pack[(pipe.n_submitted - 1) % args.depth]  # this slot's staging buffer
                for b in range(s1 - s0):
                    pack[b].copy_(slot.renders_tex[b] if args.with_color else slot.renders[b])
                gather.push(s0 // batch, pack)
            if log:
                status_log.append(slot.status[:s1 - s0].clone())  # device-side copy, no sync

    def timed_pass(log):
        """Warm-up batches, then EXACTLY --steps frames between barrier + synchronize pairs."""
        for s0 in range(0, n_warm, batch):
            run_batch(s0, min(s0 + batch, n_warm), False)
        pipe.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for s0 in range(n_warm, n_frames, batch):
            run_batch(s0, min(s0 + batch, n_frames), log)
        pipe.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0

    elapsed = timed_pass(True)

    # informational second pass (N=1 only, never `value`): the same frames with the MLP on the
    # f32-accurate f16x3 kernel, plus the largest difference between the two volumes of one frame
    alt = None
    if world == 1 and args.precision == "f32" and not args.no_alt and not args.with_color:
        last_slot = pipe.slots[(pipe.n_submitted - 1) % len(pipe.slots)]
        vol_f32 = last_slot.volumes[last_slot.n_active - 1].clone()
        head = pipe.slots[0].net.surface_classifier
        head.set_precision("f16x3")
        head.packed()  # re-pack now, on this stream, and drain before the slots' streams use it
        torch.cuda.synchronize()
        alt_elapsed = timed_pass(False)
        last_slot = pipe.slots[(pipe.n_submitted - 1) % len(pipe.slots)]
        vol_alt = last_slot.volumes[last_slot.n_active - 1]
        diff = (vol_alt - vol_f32).abs().max().item()
        flips = int(((vol_alt > 0.5) != (vol_f32 > 0.5)).sum().item())
        head.set_precision("f32")
        head.packed()
        torch.cuda.synchronize()
        alt = {"precision": "f16x3 (f32 emulated on f16 MFMA, 3-term split, f32 accumulate)",
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
    # --with-color one netC launch per frame follows, which the netG roofline skips
    launch_ms, prof_pts, cursor = [], [], 0
    for st in prof_status:
        counts = st.cpu().numpy()[:, 1:]
        for b0 in range(0, counts.shape[0], MAX_RECON_BATCH):
            prof_pts.append(counts[b0:b0 + MAX_RECON_BATCH].sum(0))
            launch_ms.append(all_ms[cursor:cursor + len(resolutions)])
            cursor += len(resolutions)
        if args.with_color:
            cursor += counts.shape[0]
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
            "metric": "reconstructions/sec (512^2 in, 256^3 grid)",
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
                "parallelism": "frame-parallel x%d; per GPU %d slots x %d frames in flight, encoder "
                               "batched per slot%s"
                               % (world, args.depth, batch,
                                  "" if args.no_graph else " and replayed as a hipGraph"),
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
                "traffic": traffic_from_profile(min(batch, MAX_RECON_BATCH)),
                "launches": int(n_launch),
                "frames_per_launch": min(batch, MAX_RECON_BATCH),
                "avg_launch_ms": float(launch_ms[:n_launch].mean()) if n_launch else None,
                "flop_per_point": FLOP_PER_POINT,
            },
        }
        if alt is not None:
            out["alt_precision"] = alt
        if world == 1 and not args.no_cpu_baseline and not args.with_color and args.levels == 5:
            # bounded thread count: torch-CPU convs at batch 1 collapse when oversubscribed
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 32))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
