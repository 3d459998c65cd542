#!/usr/bin/env python
"""Headline benchmark: reconstructions/sec (512x512 image in, 256^3-effective octree grid out).

One "step" = one full geometry reconstruction of one synthetic frame on one MI355X
(BASELINE.json configs[1]): netG.filter (hourglass encoder) -> channels-last features ->
the frame's skip table (the MLP's products with the sampled feature, once per texel) ->
coarse-to-fine octree 17..257 driving the fused HIP query kernel -> forward_vertices -> normal
render.  Inputs are resident in HBM before the timed region.  With --gpus N every rank
reconstructs its own frames (frame-parallel, weak scaling) and the renders are gathered to rank 0
over RCCL.

    python bench.py [--gpus N --steps K --warmup W]

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment launches its own N ranks (one per
GPU, ``python -m torch.distributed.run --standalone``-style on 127.0.0.1) and fails loudly when
fewer than N GPUs are visible; under ``python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N ...`` it joins the ranks the launcher made.  Either way rank 0 prints ONE JSON
line with the driver's contract fields plus

  "roofline"      the fused query kernel against the f32 MFMA peak (HIP-event timed, live): `frac` prices the
                  FLOPs it EXECUTES (`frac_definition`); `like_for_like_frac` = the plain kernel of this run, where
                  executed = the reference's FLOPs per point; `reference_flops_equivalent` = the reference's
                  per-point FLOPs over the same launch times as a rate-equivalent and a speed-up (the skip tables
                  hoist 42 % of them out of the per-point work), never as a fraction;
                  `roofline.step` = the query launches AND skip_table_kernel as one rate;
                  `roofline.traffic` = memory-side bytes per launch from the committed PMC
                  passes at this frames-per-launch (20 with --steps 20, 32 with the default 32);
                  `roofline.sustained` = what a register-only loop of the same MFMA instruction delivers on this
                  box right after the timed launches, with the shader clock it ran at (`peak` stays nominal)
  "plain_query_path"  (N=1 only) the headline configuration without skip tables (--no-skip-table)
  "cpu_baseline"  the CPU oracle path timed on this box's host cores (rank 0, N=1 only); "cpu_baseline_reference_ops" =
                  the same reconstruction with the query through the reference's own torch CPU operators, op for op
                  (oracle/torch_ops.py: what the reference's CPU recon path costs on THIS box)
  "passes"        the timed region is run 5 times (each EXACTLY --steps frames between barrier +
                  synchronize pairs); `value` is the median pass, min / max are listed
  "scaling_vs_single_rank"  (N > 1) rank 0 alone on the same frames while the others wait: efficiency inside ONE run
  "in_flight_8"   BASELINE configs[3]: 8 frames in flight across the node (8/N per rank)
  "with_color" / "levels6_f16w" / "mesh" / "dropin" / "alt_precision"   (N=1 only) the other
                  BASELINE configs and surfaces, each with its own timing and parity deltas

Test hooks for the N > 1 code on a ONE-GPU box (tests/test_dropin_gpu.py): MONOPORT_BENCH_ONE_GPU_TEST=1 (--gpus 2,
both ranks on device 0, collectives over gloo) and MONOPORT_BENCH_FORCE_GROUP=1 (--gpus 1 inside a one-rank RCCL
group: the same collective calls on the real backend).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from monoport_amd import _lib, ops, parallel, synthetic as syn  # noqa: E402
from monoport_amd.pipeline import MAX_RECON_BATCH, FramePipeline  # noqa: E402
from monoport_amd.recon import pifu_calib  # noqa: E402

from bench_common import (B_MAX, B_MIN, CPU_BASELINE_REFERENCE, F32_MFMA_PEAK_TFLOPS, FLOP_PER_POINT,  # noqa: E402,F401
                          FLOP_PER_POINT_C, FLOP_PER_POINT_SKIP_TABLE, FLOP_SKIP_TABLE_PER_FRAME, HBM_PEAK_GBS,
                          N_IMAGES, RESOLUTIONS, build_netc, build_netg, set_precision_everywhere)
from bench_dropin import dropin_surface, soak  # noqa: E402


def make_pipeline(device, depth, use_graph, resolutions=None, with_color=False, precision="f32",
                  batch=1, final_level="dilate3"):
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
                         use_graph=use_graph, netC=build_netc(device) if with_color else None,
                         final_level=final_level)
    pipe.prepare()
    return pipe


TRAFFIC_PROFILE_513_F16W = "r05_query_traffic_513_f16w.json"  # configs[4]: tools/r05_run.sh traffic16
TRAFFIC_PROFILE = "r05_query_traffic.json"  # PMC passes at slot batches of 16, 20, 24 and 32 frames (tools/r05_run.sh traffic)


def traffic_from_profile(precision, levels, with_color, slot_batch):
    """HBM-side bytes per fused-query launch from the committed PMC passes (separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE runs of tools/traffic_probe.py, corrected as
    MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside this process, so the
    figure is reported ONLY for the configurations the passes covered (f32 skip-table kernel, 5 levels,
    geometry only, slot batches of 20 / 32 / 16 / 24 frames: --steps 20, the default --steps 32, --steps 48, --batch 24)
    and is None for every other run or when the profile is absent."""
    if precision == "f16w" and levels == 6 and not with_color:  # configs[4]: its own passes (16 frames per launch)
        try:
            with open(os.path.join(ROOT, "profiles", TRAFFIC_PROFILE_513_F16W)) as f:
                prof = json.load(f)
            return prof["bytes_per_launch_avg"] if int(slot_batch) == int(prof["slot_batch"]) else None
        except (OSError, KeyError, ValueError):
            return None
    if precision != "f32" or levels != 5 or with_color:
        return None
    path = os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)
    try:
        with open(path) as f:
            prof = json.load(f)
        return prof["by_slot_batch"][str(int(slot_batch))]["bytes_per_launch_avg"]
    except (OSError, KeyError, ValueError):
        return None


# ---------------------------------------------------------------------------------------------
# launching the ranks
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """``python bench.py --gpus N`` (N > 1) outside a launcher: start N ranks of this script, one
    per GPU, with torch.distributed.run on 127.0.0.1 and pass their output / exit code through.
    Refuses (exit 2) when fewer than N GPUs are visible -- never a silent 1-GPU run."""
    one_gpu_test = os.environ.get("MONOPORT_BENCH_ONE_GPU_TEST") == "1"
    if not args.rendezvous_only:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs MI355X GPUs (no CPU fallback)")
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus and not one_gpu_test:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible on this node; "
                             "refusing to run (a frame-parallel measurement needs one GPU per "
                             "rank)\n" % (args.gpus, n_dev))
            raise SystemExit(2)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL)
    env.setdefault("OMP_NUM_THREADS", "4")
    env["MONOPORT_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
           str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("bench.py: launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def rank_devices(dist, device, world):
    """(device index, PCI bus id, name) of every rank, gathered on all ranks."""
    if device.type == "cuda":
        props = torch.cuda.get_device_properties(device)
        bus = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1),
                                  getattr(props, "pci_device_id", -1))
        mine = "%d|%s|%s" % (torch.cuda.current_device(), bus, props.name)
    else:
        mine = "cpu|pid%d" % os.getpid()
    if dist is None:
        return [mine]
    got = [None] * world
    dist.all_gather_object(got, mine)
    return got


def rendezvous_only(args):
    """--rendezvous-only: every rank joins the process group, the device census and ONE gather of
    a frame-sized payload run exactly as in a measurement, and rank 0 prints a JSON line -- no
    reconstruction, no GPU kernels of ours.  It proves the launch path (self-launch or
    torch.distributed.run) without spending a measurement, and runs on a box without GPUs (gloo;
    tests/test_bench_launch_cpu.py)."""
    have_gpu = torch.cuda.is_available()
    one_gpu_test = os.environ.get("MONOPORT_BENCH_ONE_GPU_TEST") == "1"
    local_rank = 0 if one_gpu_test else int(os.environ.get("LOCAL_RANK", "0"))
    if have_gpu:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
    backend = "nccl" if have_gpu and not one_gpu_test else "gloo"
    rank, world = parallel.init_from_env(backend=backend, device=device if have_gpu else None)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
    devices = rank_devices(dist, device, world)
    if world > 1 and have_gpu and not one_gpu_test:
        assert len(set(devices)) == world, "ranks share a GPU: %s" % devices
    gather = parallel.FrameGather((1, 257, 257, 3), device=device, store=False)
    payload = torch.full((1, 257, 257, 3), float(rank), device=device)
    t_g = time.perf_counter()
    gather.push(0, payload)
    if have_gpu:
        torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t_g) * 1e3
    ok = True
    if rank == 0 and world > 1:
        ok = all(bool((gather.received(r) == float(r)).all()) for r in range(world))
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "backend": backend,
                          "devices": devices, "gather_checked": bool(ok), "gather_ms_first": gather_ms,
                          "self_launched": os.environ.get("MONOPORT_BENCH_SELF_LAUNCHED") == "1"}),
              flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


# ---------------------------------------------------------------------------------------------
# the measurement
# ---------------------------------------------------------------------------------------------
class Job:
    """Rank identity and the synthetic inputs every leg shares."""

    def __init__(self, device, rank, world, dist, one_gpu_test, steps, warm):
        self.device, self.rank, self.world, self.dist = device, rank, world, dist
        self.one_gpu_test = one_gpu_test
        self.steps, self.warm = steps, warm
        self.gather_ms = None  # per warm-up submission: duration of the render gather on this rank (N > 1)
        n_frames = steps + warm
        # distinct frames per rank: frame id = step * world + rank (frame-parallel sharding)
        self.images = [torch.from_numpy(syn.synthetic_image(s * world + rank))[None].to(device)
                       for s in range(min(n_frames, N_IMAGES))]
        self.calibs = [pifu_calib(*syn.scene_camera(3 * (s * world + rank)), device=device)
                       for s in range(n_frames)]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def reduce_max(self, values):
        """Element-wise MAX over ranks of a list of floats (the slowest rank defines a pass)."""
        if self.dist is None:
            return [float(v) for v in values]
        t = torch.tensor(values, dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def gather_floats(self, value):
        if self.dist is None:
            return [float(value)]
        got = [None] * self.world
        self.dist.all_gather_object(got, float(value))
        return got


def pick_batch(steps, upper, depth=3):
    """Frames per slot submission -- ONE rule for every --steps: equal submissions, as many as there are slots
    (`depth`) when --steps divides that way, else as few as possible; never more than `upper` frames each (--batch;
    MAX_RECON_BATCH = 32 = kMaxFrames of mp_recon_batch when not given), so that no submission is short (a short
    batch would still pay the full-batch encoder) and every octree level of a submission is ONE fused-query launch.
    The default 32 steps -> one submission of 32 (194-195 recon/s); the driver's 20 steps -> one submission of
    20 (190; its frames complete together: `config.frame_latency_ms`; the two-submission layout is reported as
    `two_slot_submissions`); 48 steps on 3 slots -> 3 x 16 (183-190, passes scatter: which slot's encoder meets
    which slot's octree is a matter of timing); 96 -> 3 x 32."""
    upper = MAX_RECON_BATCH if upper is None else upper
    if steps % depth == 0 and 1 <= steps // depth <= upper:
        return steps // depth
    return max(b for b in range(1, max(1, min(upper, steps)) + 1) if steps % b == 0)


def timed_passes(job, pipe, batch, with_color, passes, collective=True, want_status=True):
    """Warm-up, then `passes` timed regions of EXACTLY job.steps frames each, every one between
    barrier + synchronize pairs; returns (per-pass seconds of THIS rank, status rows of the first
    pass, gather_checked).  `collective=False` runs the same frames without the render gather and
    the barriers (the single-rank leg of an N > 1 run)."""
    gathering = collective and job.dist is not None  # N > 1 (or the forced one-rank group of the test hook)
    depth = len(pipe.slots)
    r_last = pipe.slots[0].res[-1]
    n_warm, n_frames = job.warm, job.steps + job.warm
    gather = parallel.FrameGather((batch, r_last, r_last, 3), device=job.device, store=False) \
        if gathering else None
    render_pack = [torch.zeros((batch, r_last, r_last, 3), dtype=torch.float32, device=job.device)
                   for _ in range(depth)] if gathering else None
    gather_checked = [False]
    gather_events = []  # (start, stop) HIP events around the warm-up gathers, on the slot's stream
    warming = [True]
    status_log = []

    def run_batch(s0, s1, log):
        """Frames s0 .. s1-1 (at most `batch`) as one slot submission."""
        slot = pipe.submit([job.images[s % len(job.images)] for s in range(s0, s1)],
                           [job.calibs[s] for s in range(s0, s1)])
        with torch.cuda.stream(slot.stream):
            if gathering:
                pack = render_pack[(pipe.n_submitted - 1) % depth]  # this slot's staging buffer
                for b in range(s1 - s0):
                    pack[b].copy_(slot.renders_tex[b] if with_color else slot.renders[b])
                ev = None
                if warming[0]:  # warm-up submissions: time the collective (events on the stream it is enqueued on)
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(slot.stream)
                gather.push(s0 // batch, pack)
                if ev is not None:
                    ev[1].record(slot.stream)
                    gather_events.append(ev)
                if not log and job.rank == 0 and not gather_checked[0]:
                    # (warm-up only: this syncs) the gathered copy of rank 0's own frames must
                    # equal what rank 0 rendered
                    got = gather.received(0)[:s1 - s0].to(pack.device)
                    assert torch.equal(got, pack[:s1 - s0]), "gather mismatch"
                    gather_checked[0] = True
            if log:
                status_log.append(slot.status[:s1 - s0].clone())  # device-side copy, no sync

    def bracket():
        pipe.synchronize()
        torch.cuda.synchronize()
        if collective:
            job.barrier()

    for s0 in range(0, n_warm, batch):
        run_batch(s0, min(s0 + batch, n_warm), False)
    warming[0] = False
    if gather_events:
        bracket()
        job.gather_ms = [a.elapsed_time(b) for a, b in gather_events]
    elapsed = []
    for p in range(passes):
        bracket()
        t0 = time.perf_counter()
        for s0 in range(n_warm, n_frames, batch):
            run_batch(s0, min(s0 + batch, n_frames), want_status and p == 0)
        bracket()
        elapsed.append(time.perf_counter() - t0)
    statuses = torch.cat(status_log).cpu().numpy() if status_log else None
    return elapsed, statuses, gather_checked[0]


def roofline_leg(job, pipe, batch, resolutions, with_color):
    """The same frames again on ONE stream with every fused-query launch bracketed by HIP events on
    its launch stream (concurrent slots would share CUs) -> per-launch durations and point counts
    of the dominant kernel (and of the netC colour launches with `with_color`)."""
    n_warm, n_frames = job.warm, job.steps + job.warm
    slot = pipe.slots[0]
    prof_status, prof_vcount = [], []
    table_ms = None
    if slot.tables is not None:  # the skip tables of one slot submission (skip_table_kernel), per frame
        mlp = slot.net.surface_classifier.packed()
        nb = min(batch, slot.feat_hwc_all.shape[0])
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(slot.stream):
            for rep in range(6):
                if rep == 1:
                    ev0.record(slot.stream)
                handle = ops.skip_table_batch(mlp, slot.feat_hwc_all[:nb], out=slot.tables[:nb])
            ev1.record(slot.stream)
        slot.stream.synchronize()
        slot._table_handle = handle
        table_ms = ev0.elapsed_time(ev1) / 5 / nb
    cap = 8 * job.steps + 64
    ops.profile_begin(job.device, max_records=cap)
    for s0 in range(n_warm, n_frames, batch):
        s1 = min(s0 + batch, n_frames)
        slot.submit([job.images[s % len(job.images)] for s in range(s0, s1)],
                    [job.calibs[s] for s in range(s0, s1)])
        with torch.cuda.stream(slot.stream):
            prof_status.append(slot.status[:s1 - s0].clone())
            if with_color:
                prof_vcount.append(torch.cat([slot.vertices[b][4].reshape(1) for b in range(s1 - s0)]))
    slot.wait()
    all_ms = ops.profile_end(job.device, capacity=cap)
    # launch order per submission of n frames: for every chunk of <= MAX_RECON_BATCH (32) frames one launch per level
    # (its points = that level's nodes summed over the chunk, pipeline.py / mp_recon_batch); with
    # colour one netC launch per chunk follows the octree launches of the whole submission
    launch_ms, launch_pts, c_ms, c_pts, cursor = [], [], [], [], 0
    for i, st in enumerate(prof_status):
        counts = st.cpu().numpy()[:, 1:]
        for b0 in range(0, counts.shape[0], MAX_RECON_BATCH):
            launch_pts.append(counts[b0:b0 + MAX_RECON_BATCH].sum(0))
            launch_ms.append(all_ms[cursor:cursor + len(resolutions)])
            cursor += len(resolutions)
        if with_color:
            vc = prof_vcount[i].cpu().numpy()
            for b0 in range(0, counts.shape[0], MAX_RECON_BATCH):
                c_pts.append(int(vc[b0:b0 + MAX_RECON_BATCH].sum()))
                c_ms.append(float(all_ms[cursor]))
                cursor += 1
    launch_ms = np.concatenate(launch_ms)
    launch_pts = np.stack(launch_pts)
    n_launch = min(len(launch_ms), launch_pts.size)
    flops = launch_pts.reshape(-1)[:n_launch].astype(np.float64) * FLOP_PER_POINT
    achieved = flops.sum() / (launch_ms[:n_launch].sum() * 1e-3) / 1e12 if n_launch else 0.0
    out = {"achieved": achieved, "launches": int(n_launch), "launch_ms": launch_ms[:n_launch],
           "launch_pts": launch_pts.reshape(-1)[:n_launch], "points_per_level": launch_pts.sum(0),
           "skip_table_ms_per_frame": table_ms, "frames": int(sum(st.shape[0] for st in prof_status))}
    if with_color and c_ms:
        out["color_achieved"] = (np.sum(c_pts, dtype=np.float64) * FLOP_PER_POINT_C
                                 / (np.sum(c_ms) * 1e-3) / 1e12)
        out["color_points_per_frame"] = float(np.sum(c_pts)) / job.steps
    return out


def roofline_step(roof, skip_on, peak_tflops):
    """ALL kernels that produce the field of a frame -- the fused-query launches and, with skip tables,
    skip_table_kernel -- as one rate: `executed` (what the kernels multiply: 1,380,354 FLOP per point + 16.1
    GFLOP per frame of tables; a roofline fraction) and `reference_flops_equivalent` (the reference's 2,363,906
    FLOP per point, SURVEY 8d, nothing credited for the tables; a speed-up, not a fraction)."""
    if not roof["launches"] or not roof.get("frames"):
        return None
    q_ms = float(roof["launch_ms"].sum())
    pts = float(roof["launch_pts"].sum())
    frames = roof["frames"]
    t_ms = (roof["skip_table_ms_per_frame"] or 0.0) * frames if skip_on else 0.0
    ex_flop = pts * (FLOP_PER_POINT_SKIP_TABLE if skip_on else FLOP_PER_POINT) + (FLOP_SKIP_TABLE_PER_FRAME * frames if skip_on else 0)
    al_flop = pts * FLOP_PER_POINT
    sec = (q_ms + t_ms) * 1e-3
    return {"kernels": "fused-query launches" + (" + skip_table_kernel" if skip_on else ""),
            "query_ms_per_frame": q_ms / frames, "skip_table_ms_per_frame": t_ms / frames if skip_on else None,
            "executed": {"tflops": ex_flop / sec / 1e12, "frac": ex_flop / sec / 1e12 / peak_tflops},
            "reference_flops_equivalent": {"tflops_equivalent": al_flop / sec / 1e12,
                                           "speedup_vs_reference_flops_at_peak": al_flop / sec / 1e12 / peak_tflops}}


def breakdown_leg(job, pipe, batch, resolutions):
    """encoder-only and encoder-excluded time per frame, one stream."""
    slot = pipe.slots[0]
    r_last = resolutions[-1]

    def timed(fn, n):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(slot.stream):
            fn()
            ev0.record(slot.stream)
            for _ in range(n):
                fn()
            ev1.record(slot.stream)
        slot.stream.synchronize()
        return ev0.elapsed_time(ev1) / n

    def recon_only():
        mlp = slot.net.surface_classifier.packed()
        ops.recon(mlp, slot.feats_hwc[0], slot.calib[0:1], syn.Z_SCALE, B_MIN, B_MAX,
                  resolutions, 0.5, volume=slot.volume, status=slot.status[0])
        x, y, z, nrm, count = ops.forward_vertices_raw(slot.volume, "front")
        ops.paint(x, y, nrm, 0, count, r_last, 0.5, 0.5, 0.0, 1.0)

    def recon_batched():
        """What a slot does after its encoder: the octree of its frames level by level, then per
        frame forward_vertices + render (monoport_amd/pipeline.py)."""
        mlp = slot.net.surface_classifier.packed()
        nb = min(batch, MAX_RECON_BATCH)
        ops.recon_batch(mlp, slot.feats_hwc[:nb], slot.calib[:nb], syn.Z_SCALE, B_MIN, B_MAX,
                        resolutions, 0.5, volumes=slot.volumes[:nb], status=slot.status[:nb])
        raws = ops.forward_vertices_raw_batch(slot.volumes[:nb], "front")
        ops.paint_batch([v[0] for v in raws], [v[1] for v in raws], [v[3] for v in raws], 0, [v[4] for v in raws],
                        r_last, 0.5, 0.5, 0.0, 1.0)

    with torch.no_grad():
        # the last submission may have been a short batch: refill the slot so every entry is live
        slot.submit([job.images[s % len(job.images)] for s in range(batch)], job.calibs[:batch])
        slot.wait()
        enc_ms = timed(lambda: slot.net.image_filter(slot.image, last_only=True), 10) / batch
        # ... and as the timed region runs it: the slot's captured hipGraph (static buffers; the eager figure moves
        # with the state of torch's caching allocator this late in the process: 1.77 or 1.87 ms on the same code)
        enc_graph_ms = timed(slot.graph.replay, 10) / batch if slot.graph is not None else None
        enc1_ms = timed(lambda: slot.net.image_filter(slot.image[:1], last_only=True), 10)
        # the same passes with every 3x3 convolution on the direct implicit-GEMM kernels (mp_conv3x3_tune(0x400): the
        # launcher ignores the Winograd-domain weights) -- what csrc/conv_wino.hip buys, measured in this process
        lib = _lib.load()
        lib.mp_conv3x3_tune(0x400)
        try:
            enc_direct_ms = timed(lambda: slot.net.image_filter(slot.image, last_only=True), 10) / batch
            enc1_direct_ms = timed(lambda: slot.net.image_filter(slot.image[:1], last_only=True), 10)
        finally:
            lib.mp_conv3x3_tune(0)
        rec_ms = timed(recon_only, 10)
        rec_batched_ms = timed(recon_batched, 5) / min(batch, MAX_RECON_BATCH)
    return {
        "encoder_ms_per_frame": enc_ms, "encoder_ms_batch1": enc1_ms,
        "encoder_ms_per_frame_as_run": enc_graph_ms,
        "encoder_ms_per_frame_direct_conv3x3": enc_direct_ms, "encoder_ms_batch1_direct_conv3x3": enc1_direct_ms,
        "recon_vertices_render_ms": rec_ms,
        "recon_vertices_render_ms_per_frame_batched": rec_batched_ms,
        "recon_per_s_encoder_excluded": 1e3 / rec_batched_ms,
        "recon_per_s_encoder_excluded_single_frame": 1e3 / rec_ms,
        "encoder_conv3x3": ("Winograd F(2x2,3x3) on f32 MFMA for the launches csrc/conv_wino.hip serves (Cout % 64 == 0, >= 128 "
                            "workgroups: 4/9 of the direct form's multiplies, exact-f32 products), direct implicit GEMM otherwise"
                            if ops.CONV_WINOGRAD else "direct implicit GEMM on f32 MFMA (MONOPORT_CONV_WINOGRAD=0)"),
        "note": "single stream, no overlap; encoder eager at the bench batch size (and at batch 1), `as_run` = the hipGraph of "
                "the slot (with --with-color: both encoders); "
                "batched = mp_recon_batch over the slot's frames, as the pipeline runs it",
    }


def mesh_leg(job, volume, resolutions):
    """North star's mesh output: marching cubes (csrc/mcubes.hip) of one reconstructed volume at the
    full resolution, HIP-event timed.  HBM-bound; algorithmic bytes = one read of the volume + 12 B
    per vertex + 12 B per face.  SELF-PARITY: the reference has no marching cubes, connectivity is
    checked against our own CPU oracle (tests/test_recon_gpu.py, 257^3 included)."""
    r = resolutions[-1]
    torch.cuda.synchronize()
    verts, faces, counts = ops.marching_cubes_raw(volume, 0.5, B_MIN, B_MAX)
    nv, nf = (int(c) for c in counts.cpu())
    if nv > verts.shape[0] or nf > faces.shape[0]:
        return {"error": "capacity guess too small (%d verts, %d faces)" % (nv, nf)}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    ev0.record()
    for _ in range(reps):
        ops.marching_cubes_raw(volume, 0.5, B_MIN, B_MAX)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    alg_bytes = 4.0 * r ** 3 + 12.0 * nv + 12.0 * nf
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    f = faces[:nf].long()
    return {
        "kernel": "marching cubes: mc_count / mc_scan / mc_vertices / mc_triangles (csrc/mcubes.hip)",
        "resolution": r, "mesh_ms": ms, "vertices": nv, "faces": nf,
        "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes},
        "euler_characteristic": int(nv - torch.unique(torch.sort(torch.cat(
            [f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1).values, dim=0).shape[0] + nf),
        "parity": "self-parity: identical connectivity to our CPU oracle (the reference has no "
                  "marching cubes, SURVEY.md section 0)",
    }


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
    # the same reconstruction with the query through the reference's OWN torch CPU operators (oracle/torch_ops.py:
    # baddbmm, grid_sample, the Conv1d chain, op for op -- bit-identical to the reference-generated goldens where those
    # were made): the encoder above already is the reference's op set, so this is what "the reference CPU recon path"
    # costs on THIS box's host cores (/root/reference itself does not exist here)
    from oracle import torch_ops
    stats_t = []
    t4 = time.perf_counter()
    vol_t = orc.seg3d_lossless(lambda p: torch_ops.query(feat, p, calib, layers, 1, syn.Z_SCALE)[0],
                               B_MIN, B_MAX, RESOLUTIONS, stats=stats_t)
    t5 = time.perf_counter()
    total_t = (t1 - t0) + (t5 - t4) + (t3 - t2)
    ref_ops = {
        "value": 1.0 / total_t, "unit": "recon/s", "cores": threads, "kind": "port",
        "ops": "the reference's torch CPU operators restated op for op (MonoPortNet.py:48-91 -> oracle/torch_ops.py)",
        "sample": "1 reconstruction: encoder %.2fs (torch CPU) + octree %.2fs (%d pts through baddbmm / grid_sample / "
                  "Conv1d, %d torch threads) + forward_vertices %.2fs" % (t1 - t0, t5 - t4, sum(stats_t), threads, t3 - t2),
        "mpts_per_s": sum(stats_t) / (t5 - t4) / 1e6,
        "max_abs_diff_vs_c_oracle_volume": float(np.abs(vol_t - vol).max()),
    }
    return {
        "value": 1.0 / total, "unit": "recon/s", "cores": threads, "kind": "port",
        "sample": "1 reconstruction: encoder %.2fs (torch CPU) + octree %.2fs (%d pts, C oracle f32, "
                  "OpenMP) + forward_vertices %.2fs" % (t1 - t0, t2 - t1, sum(stats), t3 - t2),
        "mpts_per_s": sum(stats) / (t2 - t1) / 1e6,
        "reference_ops": ref_ops,
    }


def in_flight_layout(k_total, world):
    """--in-flight K: K frames in flight across the node = K / N per rank, as (slots, frames per
    slot).  One frame per rank (configs[3] on 8 GPUs) is one slot of one frame; an even share is
    split over two slots so that one slot's encoder overlaps the other's octree."""
    if k_total % world != 0:
        raise SystemExit("--in-flight %d is not a multiple of --gpus %d" % (k_total, world))
    per_rank = k_total // world
    depth = 2 if per_rank >= 2 and per_rank % 2 == 0 else 1
    return depth, per_rank // depth


def build_pipeline(job, depth, batch, use_graph, resolutions, with_color, precision, final_level="dilate3"):
    """make_pipeline on every rank, agreed: if hipGraph capture fails on ANY rank (it can next to an
    initialised RCCL communicator, whose watchdog thread touches the runtime) all ranks rebuild
    with eager encoder launches rather than lose the run.  Returns (pipeline, use_graph)."""
    pipe, ok = None, 1
    try:
        pipe = make_pipeline(job.device, depth, use_graph, resolutions, with_color, precision, batch, final_level)
    except RuntimeError as e:
        if not use_graph:
            raise
        sys.stderr.write("bench: hipGraph capture failed (%s); falling back to --no-graph\n" % e)
        torch.cuda.synchronize()
        ok = 0
    if job.dist is not None:  # all ranks run the same variant
        flag = torch.tensor([ok], device=job.device)
        job.dist.all_reduce(flag, op=job.dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok:
        if pipe is not None:
            pipe.close()
        use_graph = False
        pipe = make_pipeline(job.device, depth, False, resolutions, with_color, precision, batch, final_level)
    return pipe, use_graph


def measure_config(job, depth, batch, use_graph, resolutions, with_color, precision, passes,
                   roofline=True, final_level="dilate3"):
    """Build a pipeline for one configuration, run the timed passes (+ the roofline leg), return
    (summary dict, pipeline).  The caller closes the pipeline."""
    pipe, use_graph = build_pipeline(job, depth, batch, use_graph, resolutions, with_color, precision, final_level)
    elapsed, statuses, checked = timed_passes(job, pipe, batch, with_color, passes)
    # a pass is as slow as its slowest rank; the median pass is the reported one
    per_pass = job.reduce_max(elapsed)
    order = sorted(per_pass)
    med = order[len(order) // 2]
    pts = float(statuses[:, 1:].sum()) if statuses is not None else 0.0
    ok = bool((statuses[:, 0] == 1).all()) if statuses is not None else True
    if job.dist is not None:
        t = torch.tensor([pts, float(ok)], dtype=torch.float64, device=job.device)
        job.dist.all_reduce(t, op=job.dist.ReduceOp.SUM)
        pts, ok = float(t[0].item()), int(t[1].item()) == job.world
    assert ok, "synthetic body must be non-empty"
    res = {
        "value": job.steps * job.world / med, "ms_per_step": med / job.steps * 1e3,
        "elapsed": med, "points": pts, "gather_checked": checked, "use_graph": use_graph,
        "passes": {"n": passes, "value_min": job.steps * job.world / order[-1],
                   "value_max": job.steps * job.world / order[0],
                   "ms_per_step_all": [e / job.steps * 1e3 for e in per_pass]},
        "points_per_level": (statuses[:, 1:].sum(0) / float(statuses.shape[0])).tolist() if statuses is not None else None,
        "ms_per_step_per_rank": [e / job.steps * 1e3
                                 for e in job.gather_floats(sorted(elapsed)[len(elapsed) // 2])],
    }
    if roofline:
        res["roof"] = roofline_leg(job, pipe, batch, resolutions, with_color)
    return res, pipe


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32,
                    help="frames in the timed region (default 32 = ONE slot submission of 32 frames, every octree level one "
                         "launch of kMaxFrames frames: 194-195 recon/s, passes within 0.5 %%; the driver's 20 = one "
                         "submission of 20: 190; several overlapping submissions -- 48 = 3 x 16, 64 = 2 x 32 -- are no "
                         "faster since round 5 and their passes scatter by +- 4 %%)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--depth", type=int, default=3, help="pipeline slots (streams) per GPU")
    ap.add_argument("--batch", type=int, default=None,
                    help="frames per slot (upper bound): their encoder passes run as one batch and "
                         "their octree levels as fused-query launches of <= 32 frames; up to depth x batch frames "
                         "are in flight.  The largest divisor of --steps not above this is used, so no slot "
                         "submission is short (a short batch would still pay the full-batch encoder).  Not "
                         "given: --steps / --depth when that divides, else the largest divisor <= 32 (pick_batch)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="K > 0: exactly K frames in flight across the node (K / --gpus per rank; "
                         "BASELINE configs[3] is K = 8) instead of --depth x --batch per GPU; the "
                         "default run reports this configuration as `in_flight_8` next to `value`")
    ap.add_argument("--passes", type=int, default=5,
                    help="timed regions of exactly --steps frames each; `value` is the median (5: with three slots "
                         "overlapping, single passes of the default run differ by +- 4 %%; the driver's one-submission "
                         "run repeats to 0.1 %%)")
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
    ap.add_argument("--soak", type=float, default=10.0,
                    help="seconds of the per-frame drop-in pipeline's soak leg (dropin.soak: latency distribution and "
                         "held memory per 10-s window; `--mode dropin --soak 60` for the long form, 0 = off)")
    ap.add_argument("--final-level", default="dilate3", choices=["dilate3", "upstream", "interpolate"],
                    help="selection rule of the LAST octree level (Seg3dLossless(final_level=...)): dilate3 = the "
                         "lossless schedule (default, the headline); upstream = nodes whose upsampled mask is exactly "
                         "0.5, undilated (the rule recalled from the un-vendored implicit_seg package); interpolate = "
                         "no evaluation at the last level.  The default run reports the other two as "
                         "`final_level_rules`")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline measurement + roofline (skips every leg below)")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the drop-in-surface pass a default N=1 run appends")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-skip-table", action="store_true",
                    help="plain query path: layer 0 on the MFMAs for every point instead of the per-frame "
                         "layer-0 tables (mp_skip_table)")
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the informational f16x3 pass that a default N=1 run appends")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the with_color / levels6_f16w / in_flight_8 / mesh legs")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="join the ranks, census the devices, one frame-sized gather, print JSON; no "
                         "measurement (launch-path check; runs without GPUs over gloo)")
    return ap.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, argv)  # does not return
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ.get("WORLD_SIZE")))
    if args.rendezvous_only:
        return rendezvous_only(args)
    if args.no_extras:
        args.no_dropin = args.no_cpu_baseline = args.no_alt = args.no_configs = True
    if args.no_skip_table:
        ops.SKIP_TABLE = False
    # whether the fused query of this run blends table rows (f32 and f16x3 heads; ops.table_precision mirrors the C side)
    skip_on = ops.SKIP_TABLE and ops.table_precision(args.precision)

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/test_dropin_gpu.py): exercise the N > 1 code path on a ONE-GPU box -- every
    # rank on device 0, collectives over gloo instead of RCCL (which refuses two ranks per device)
    one_gpu_test = os.environ.get("MONOPORT_BENCH_ONE_GPU_TEST") == "1"
    if one_gpu_test:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank with LOCAL_RANK=%d but only %d GPU(s) visible"
                         % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # host cores next to this rank's GPU (N > 1; MONOPORT_BENCH_PIN=1 forces it for one rank, =0 switches it off)
    pin = os.environ.get("MONOPORT_BENCH_PIN", "auto")
    cpu_affinity = (parallel.pin_to_gpu_numa(local_rank, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", args.gpus)))
                    if pin == "1" or (pin == "auto" and args.gpus > 1 and not one_gpu_test) else
                    {"source": "unchanged (single process)", "cpus": len(os.sched_getaffinity(0))})
    backend = "gloo" if one_gpu_test else "nccl"  # nccl = RCCL on ROCm
    # second test hook: a ONE-rank RCCL group, so that a one-GPU box runs every collective call of the N > 1
    # path (barrier, all_reduce, all_gather_object, the render gather) on the real backend
    force_group = os.environ.get("MONOPORT_BENCH_FORCE_GROUP") == "1" and args.gpus == 1
    rank, world = parallel.init_from_env(backend=backend, device=device, force=force_group)
    dist = None
    if world > 1 or force_group:
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
        res = dropin_surface(device, args.steps, args.warmup, resolutions, args.passes)
        if args.soak > 0:
            res["soak"] = soak(device, args.soak, resolutions)
        print(json.dumps({
            "metric": "reconstructions/sec (512^2 in, %d^3 grid) through the drop-in surface" % (resolutions[-1] - 1),
            "value": res["value"], "unit": "recon/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": res["surface"]}, "passes": res["passes"],
            "per_frame_stages": res["per_frame_stages"], "per_frame_stages_trusted": res["per_frame_stages_trusted"],
            "latency_ms_single_frame": res["latency_ms_single_frame"], "soak": res.get("soak")}), flush=True)
        return

    job = Job(device, rank, world, dist, one_gpu_test, args.steps, args.warmup)
    if args.in_flight > 0:
        depth, batch = in_flight_layout(args.in_flight, world)
        if args.steps % batch != 0:
            raise SystemExit("--steps %d is not a multiple of the %d frames per slot that "
                             "--in-flight %d gives" % (args.steps, batch, args.in_flight))
    else:
        depth, batch = args.depth, pick_batch(args.steps, args.batch, args.depth)
    main_res, pipe = measure_config(job, depth, batch, not args.no_graph, resolutions,
                                    args.with_color, args.precision, args.passes, final_level=args.final_level)
    use_graph = main_res["use_graph"]
    roof = main_res["roof"]
    # what the f32 matrix pipe of THIS box holds right now, and at which clock (mp_mfma_clock_probe, a 30-50 ms
    # register-only MFMA loop straight after the roofline leg): boxes of one pool differ by a few per cent
    # On N > 1 EVERY rank probes its own GPU at the same time (all eight under matrix load, as in the timed region)
    # and rank 0 prints all of them: a slow or down-clocked GPU is visible in the one line.
    sustained = sustained_per_rank = None
    if args.precision == "f32":
        torch.cuda.synchronize()
        if dist is not None:
            job.barrier()
        sustained = ops.mfma_clock_probe(device, 50.0)
        if dist is not None:
            got = [None] * world
            dist.all_gather_object(got, {"rank": rank, "mfma_f32_tflops": sustained["tflops"],
                                         "shader_clock_mhz": sustained["shader_clock_mhz"]})
            sustained_per_rank = got
    r_last = resolutions[-1]
    extras = {}

    # single-rank leg of an N > 1 run: rank 0 repeats the passes alone (no gather, no barriers)
    # while the other ranks wait -> scaling efficiency against a line measured in THIS run
    if dist is not None:
        job.barrier()
        if rank == 0:
            el, _, _ = timed_passes(job, pipe, batch, args.with_color, args.passes, collective=False,
                                    want_status=False)
            single = args.steps / sorted(el)[len(el) // 2]
            extras["scaling_vs_single_rank"] = {
                "single_rank_value": single, "efficiency": main_res["value"] / (world * single),
                "note": "rank 0 alone on its GPU, same frames, other ranks idle at a barrier; the "
                        "driver computes its own efficiency from separate --gpus 1 runs"}
        job.barrier()

    # the same configuration on the PLAIN query kernels (every FLOP of the reference's MLP per point):
    # what the skip tables buy, on the driver's own record
    if world == 1 and skip_on and not args.no_configs and not args.with_color and args.levels == 5 \
            and args.in_flight == 0:
        ops.SKIP_TABLE = False
        try:
            rp, pp = measure_config(job, depth, batch, use_graph, resolutions, False, "f32", args.passes)
        finally:
            ops.SKIP_TABLE = True
        pp.close()
        del pp
        extras["plain_query_path"] = {
            "config": "the headline configuration with --no-skip-table: layer 0 and the skip connections on the MFMAs "
                      "for every point (pifu_query_kernel / pifu_query_t32_kernel)",
            "value": rp["value"], "unit": "recon/s", "ms_per_step": rp["ms_per_step"], "passes": rp["passes"],
            "roofline_frac": rp["roof"]["achieved"] / F32_MFMA_PEAK_TFLOPS,
            "roofline_achieved_tflops": rp["roof"]["achieved"], "flop_per_point": FLOP_PER_POINT}

    # the other selection rules of the last octree level (the un-vendored upstream engine's is unpinned: SURVEY 5.7)
    if world == 1 and not args.no_configs and not args.with_color and args.levels == 5 and args.in_flight == 0 \
            and args.precision == "f32" and args.final_level == "dilate3":
        rules = {}
        last_slot = pipe.slots[(pipe.n_submitted - 1) % len(pipe.slots)]
        vol_ref = last_slot.volumes[last_slot.n_active - 1].clone()  # frame steps + warm - 1, lossless schedule
        for rule in ("upstream", "interpolate"):
            rr, pr = measure_config(job, depth, batch, use_graph, resolutions, False, "f32", args.passes,
                                    roofline=False, final_level=rule)
            ls = pr.slots[(pr.n_submitted - 1) % len(pr.slots)]
            v = ls.volumes[ls.n_active - 1]
            inside = vol_ref > 0.5
            rules[rule] = {"value": rr["value"], "unit": "recon/s", "ms_per_step": rr["ms_per_step"], "passes": rr["passes"],
                           "points_per_level": rr["points_per_level"], "points_per_recon": rr["points"] / args.steps,
                           "thresholded_voxels_differing_from_dilate3": int(((v > 0.5) != inside).sum().item()),
                           "inside_voxels": int(inside.sum().item()),
                           "iou_vs_dilate3": float(((v > 0.5) & inside).sum().item()) / max(float(((v > 0.5) | inside).sum().item()), 1.0)}
            pr.close()
            del pr, ls, v
        extras["final_level_rules"] = {
            "headline_rule": "dilate3 (boundary nodes dilated by 3^3 at the last level as at levels >= 3: thresholded "
                             "volume == thresholded dense evaluation; tests/test_recon_gpu.py)",
            "headline_points_per_level": main_res["points_per_level"],
            "upstream": {"rule": "nodes whose upsampled inside-mask is exactly 0.5, undilated (`valid == 0.5`, the rule "
                                 "recalled from the upstream package's faster mode)", **rules["upstream"]},
            "interpolate": {"rule": "no evaluation at the last level (trilinear upsample of the 129^3 volume)",
                            **rules["interpolate"]},
            "note": "Seg3dLossless(final_level=...) / mp_recon_batch_ex; unpinned by the reference (implicit_seg is not "
                    "vendored): a maintainer picks the rule that reproduces their installed package"}
        del vol_ref, last_slot

    # the same frames as TWO slot submissions of steps / 2 frames on two streams (frames complete in two groups,
    # the second group's encoder under the first group's octree): the pipelined counterpart of a one-submission run
    if world == 1 and not args.no_configs and not args.with_color and args.levels == 5 and args.in_flight == 0 \
            and args.precision == "f32" and args.steps // batch == 1 and args.steps % 2 == 0 and depth >= 2:
        r2, p2 = measure_config(job, depth, args.steps // 2, use_graph, resolutions, False, "f32", args.passes,
                                roofline=False, final_level=args.final_level)
        p2.close()
        del p2
        extras["two_slot_submissions"] = {
            "config": "the headline frames as 2 submissions of %d frames on two streams" % (args.steps // 2),
            "value": r2["value"], "unit": "recon/s", "ms_per_step": r2["ms_per_step"], "passes": r2["passes"]}

    # BASELINE configs[3]: 8 frames in flight across the node, at every N that divides 8
    if not args.no_configs and args.in_flight == 0 and 8 % world == 0 and not args.with_color \
            and args.precision == "f32" and args.levels == 5:
        d8, b8 = in_flight_layout(8, world)
        if args.steps % b8 == 0:
            r8, p8 = measure_config(job, d8, b8, use_graph, resolutions, False, "f32", args.passes,
                                    roofline=False)
            p8.close()
            del p8
            extras["in_flight_8"] = {
                "config": "BASELINE configs[3]: 8 frames in flight across %d GPU(s) = %d slot(s) x %d "
                          "frame(s) per rank" % (world, d8, b8),
                "value": r8["value"], "unit": "recon/s", "ms_per_step": r8["ms_per_step"],
                "passes": r8["passes"], "ms_per_step_per_rank": r8["ms_per_step_per_rank"]}

    alt = None
    if world == 1 and args.precision == "f32" and not args.no_alt and not args.with_color:
        # informational second configuration (never `value`): the same frames with the MLP on the
        # f32-accurate f16x3 kernel, plus the largest difference between the two volumes of one frame
        last_slot = pipe.slots[(pipe.n_submitted - 1) % len(pipe.slots)]
        vol_f32 = last_slot.volumes[last_slot.n_active - 1].clone()
        # a SECOND pipeline (own network copy, own captured graphs): the f32 pipeline and its graphs
        # stay untouched for the breakdown leg below
        # slots of <= 16 frames here even for a short run: with the f16 kernels a frame has more non-MFMA time,
        # and two slots overlapping each other beat one long submission (290 vs 321 recon/s at --steps 20)
        r16, pipe16 = measure_config(job, depth, pick_batch(args.steps, args.batch or 16), use_graph, resolutions, False, "f16x3",
                                     args.passes, roofline=False)
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
               "value": r16["value"], "unit": "recon/s", "ms_per_step": r16["ms_per_step"],
               "passes": r16["passes"],
               "max_abs_diff_vs_f32_volume": diff, "thresholded_voxels_differing": flips,
               "voxels": int(vol_f32.numel()),
               "note": "opt-in (--precision f16x3); not the headline"}

    breakdown = None
    if rank == 0 or world == 1:
        breakdown = breakdown_leg(job, pipe, batch, resolutions)
        breakdown["points_per_level"] = [float(v) / args.steps for v in roof["points_per_level"]]

    if world == 1 and not args.no_configs and not args.with_color and args.levels == 5:
        slot = pipe.slots[0]
        extras["mesh"] = mesh_leg(job, slot.volumes[0], resolutions)

    if world == 1 and not args.no_configs and not args.with_color and args.levels == 5 \
            and args.precision == "f32" and args.in_flight == 0:
        # BASELINE configs[2]: geometry + netC colour
        rc, pc = measure_config(job, depth, batch, use_graph, resolutions, True, "f32", args.passes)
        pc.close()
        del pc
        extras["with_color"] = {
            "config": "BASELINE configs[2]: netG + netC (ResNet encoder + per-vertex colour MLP)",
            "value": rc["value"], "unit": "recon/s", "ms_per_step": rc["ms_per_step"],
            "passes": rc["passes"],
            "roofline_frac_netG_query": (rc["roof"]["achieved"] / F32_MFMA_PEAK_TFLOPS
                                         * (FLOP_PER_POINT_SKIP_TABLE / FLOP_PER_POINT if skip_on else 1.0)),
            "roofline_frac_netC_query": (rc["roof"].get("color_achieved", 0.0) / F32_MFMA_PEAK_TFLOPS),
            "colour_points_per_frame": rc["roof"].get("color_points_per_frame")}
        # BASELINE configs[4]: 513^3, fp16 weights; parity deltas against the exact-f32 kernel on
        # the same features and camera
        res6 = RESOLUTIONS + [513]
        r6, p6 = measure_config(job, depth, pick_batch(args.steps, args.batch or 16), use_graph, res6, False, "f16w", args.passes)
        s6 = p6.slots[0]
        s6.wait()
        mlp32 = ops.PackedMLP.from_layers(device, syn.body_mlp("G", noise=0.05, seed=1),
                                          syn.LAST_OP["G"])
        vol32, st32 = ops.recon(mlp32, s6.feats_hwc[0], s6.calib[0:1], syn.Z_SCALE, B_MIN, B_MAX, res6)
        v16 = s6.volumes[0]
        inter = ((vol32 > 0.5) & (v16 > 0.5)).sum().item()
        union = ((vol32 > 0.5) | (v16 > 0.5)).sum().item()
        extras["levels6_f16w"] = {
            "config": "BASELINE configs[4]: octree 17..513, fp16 MLP weights x split-f16 activations "
                      "(two f16 MFMAs per product, f32 accumulate)",
            "value": r6["value"], "unit": "recon/s", "ms_per_step": r6["ms_per_step"],
            "passes": r6["passes"], "points_per_recon": r6["points"] / args.steps,
            "roofline_frac": r6["roof"]["achieved"] / (2500.0 / 2), "roofline_peak_tflops": 2500.0 / 2,
            "roofline_frac_of_f16_dense_peak": r6["roof"]["achieved"] / 2500.0,
            "roofline_traffic": traffic_from_profile("f16w", 6, False, pick_batch(args.steps, args.batch or 16)),
            "roofline_traffic_source": "profiles/%s (memory side; the same file holds the L1 -> L2 request counts: the "
                                       "kernel is bound by the weight stream out of L2, 26 KB per point)" % TRAFFIC_PROFILE_513_F16W,
            "iou_vs_f32_volume": inter / max(union, 1),
            "max_abs_diff_vs_f32_volume": (vol32 - v16).abs().max().item(),
            "same_points_per_level": bool(torch.equal(st32.cpu(), s6.status[0].cpu()))}
        p6.close()
        del p6, s6, vol32, v16, mlp32

    if rank == 0:
        # the f16 variants spend 3 / 2 / 1 f16 MFMAs (2.5 PFLOP/s dense peak) per algorithmic product
        terms = {"f32": 0, "f16x3": 3, "f16w": 2, "f16": 1}[args.precision]
        peak_tflops = F32_MFMA_PEAK_TFLOPS if terms == 0 else 2500.0 / terms
        out = {
            "metric": "reconstructions/sec (512^2 in, %d^3 grid)" % (r_last - 1),
            "value": main_res["value"],
            "unit": "recon/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"],
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
                "distinct_images": len(job.images),
                "devices": devices,
                "cpu_affinity": cpu_affinity,
                # what the process group reports, not what was asked for: ranks and backend of the communicator
                "rccl_ranks": (None if dist is None else
                               {"world_size": int(dist.get_world_size()), "backend": str(dist.get_backend()),
                                "distinct_devices": len(set(devices))}),
                "backend": ("nccl (RCCL), one-rank group (test hook)" if force_group else
                            "none (single process)" if world == 1 else
                            "gloo (one-GPU test hook)" if one_gpu_test else "nccl (RCCL)"),
                "self_launched": os.environ.get("MONOPORT_BENCH_SELF_LAUNCHED") == "1",
                "gather_checked": bool(main_res["gather_checked"]) if dist is not None else None,
                # where non-linearity would come from: per-rank step times are in `ms_per_step_per_rank`; the one
                # collective of a submission (renders of `batch` frames to rank 0), timed on rank 0 during warm-up
                "gather_ms_per_submission": (None if not job.gather_ms else
                                             {"median": float(np.median(job.gather_ms)), "max": float(np.max(job.gather_ms)),
                                              "bytes_per_rank": int(batch * r_last * r_last * 3 * 4)}),
                "slot_submissions_per_rank": -(-args.steps // batch),
                "frames_in_flight_per_rank": min(depth, -(-args.steps // batch)) * batch,
                "frame_latency_ms": main_res["ms_per_step"] * batch * min(depth, -(-args.steps // batch)),
                "frame_latency_note": "a frame completes with its slot submission: up to this long after it was handed "
                                      "over (single-frame latency through the drop-in surface: latency_ms_single_frame)",
                "parallelism": "frame-parallel x%d (one process per GPU, renders gathered to rank 0 "
                               "over %s); per GPU %d slots x %d frames in flight, encoder "
                               "batched per slot%s"
                               % (world, "gloo (one-GPU test hook)" if one_gpu_test else "RCCL",
                                  depth, batch,
                                  " and replayed as a hipGraph" if use_graph else ""),
                "fixture": "F-body analytic head, seeded encoder (monoport_amd/synthetic.py)",
                "octree_schedule": "Seg3dLossless(faster=True): dilation boxes 9^3 / 7^3 / 3^3 / 3^3, last level: %s"
                                   % args.final_level,
                "points_per_level": main_res["points_per_level"],
                "points_per_recon": main_res["points"] / (args.steps * world),
            },
            "passes": main_res["passes"],
            "ms_per_step_per_rank": main_res["ms_per_step_per_rank"],
            "mpts_per_s": main_res["points"] / main_res["elapsed"] / 1e6,
            "breakdown": breakdown,
            "roofline": {
                "kernel": ("pifu_query_tabws_kernel<1> (fused MLP on 32-point tiles, wave-specialised: 4 MFMA-only consumer "
                           "waves + 4 producer waves that blend the products with the sampled feature from the frame's "
                           "skip table, skip_table_kernel)" if skip_on else
                           "pifu_query_kernel<256,1> (fused gather + MLP; launches of < 2048 tiles run on its "
                           "32-point-tile twin pifu_query_t32_kernel<1,false>)" if args.precision == "f32"
                           else "pifu_query16_tab_kernel<1,%d> (fused MLP through the skip table, %s)" % (terms, args.precision)
                           if skip_on else "pifu_query16_kernel<1,%d> (fused gather + MLP, %s)" % (terms, args.precision)),
                "bound": "mfma",
                # with the skip tables the kernel EXECUTES fewer FLOPs than the reference's MLP has
                # (`algorithmic` below): `achieved` / `frac` price the executed ones, so that
                # frac stays a statement about the kernel against the MFMA peak
                "achieved": roof["achieved"] * (FLOP_PER_POINT_SKIP_TABLE / FLOP_PER_POINT if skip_on else 1.0),
                "peak": peak_tflops,
                "unit": "TFLOP/s",
                "frac": roof["achieved"] * (FLOP_PER_POINT_SKIP_TABLE / FLOP_PER_POINT if skip_on else 1.0) / peak_tflops,
                # what `frac` prices: the FLOPs the timed kernel EXECUTES per point (PMC: SQ_INSTS_MFMA x 4096 agrees,
                # profiles/r04k_pmc_tabws.txt).  The like-for-like figure where executed = the reference's FLOPs per
                # point is the plain kernel's (`like_for_like_frac`, from the `plain_query_path` leg of this run)
                "frac_definition": "executed" if skip_on else "executed = algorithmic",
                "like_for_like_frac": (extras.get("plain_query_path", {}).get("roofline_frac") if skip_on else
                                       roof["achieved"] / peak_tflops),
                "traffic": traffic_from_profile(args.precision, args.levels, args.with_color, batch),
                "traffic_source": ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                   "this configuration)" % TRAFFIC_PROFILE),
                "launches": roof["launches"],
                "frames_per_launch": ([min(batch, MAX_RECON_BATCH)] if batch <= MAX_RECON_BATCH or batch % MAX_RECON_BATCH == 0
                                      else [MAX_RECON_BATCH, batch % MAX_RECON_BATCH]),
                "avg_launch_ms": float(roof["launch_ms"].mean()) if roof["launches"] else None,
                "flop_per_point": FLOP_PER_POINT_SKIP_TABLE if skip_on else FLOP_PER_POINT,
                # the reference's MLP per point over the same launch times is NOT a roofline fraction
                # when FLOPs are hoisted out of the launches: it is reported as a rate-equivalent and a speed-up
                "reference_flops_equivalent": {
                    "flop_per_point": FLOP_PER_POINT, "tflops_equivalent": roof["achieved"],
                    "speedup_vs_reference_flops_at_peak": roof["achieved"] / peak_tflops,
                    "note": ("a kernel running the reference's 2,363,906 FLOP per point AT the MFMA peak would be this "
                             "many times slower than these launches: 2 x 1921 x 256 FLOP per point -- every product of "
                             "weights with the sampled feature -- are hoisted out of the per-point work (a linear map "
                             "commutes with the bilinear interpolation: skip_table_kernel takes them once per texel and "
                             "frame, %d FLOP per frame, outside these launches, inside `value`); --no-skip-table runs "
                             "every FLOP per point" % FLOP_SKIP_TABLE_PER_FRAME)
                    if skip_on else "equal to the executed FLOPs"},
                "step": roofline_step(roof, skip_on, peak_tflops),
                # `peak` stays the nominal figure of MI355X_MICROARCH.md; this is what the box delivered to a
                # register-only loop of the same instruction right after the timed launches
                "sustained": None if sustained is None else {
                    "mfma_f32_tflops": sustained["tflops"], "shader_clock_mhz": sustained["shader_clock_mhz"],
                    "probe_ms": sustained["ms"],
                    "frac_of_sustained": (roof["achieved"] * (FLOP_PER_POINT_SKIP_TABLE / FLOP_PER_POINT if skip_on else 1.0)
                                          / sustained["tflops"]),
                    "per_rank": sustained_per_rank,
                    "note": "mp_mfma_clock_probe (csrc/clock_probe.hip): v_mfma_f32_32x32x2_f32 back to back on random "
                            "mantissas, two 4-wave workgroups per CU, no memory; shader clock = s_memtime cycles per "
                            "100 MHz s_memrealtime tick averaged over the workgroups"},
            },
        }
        out.update(extras)
        if alt is not None:
            out["alt_precision"] = alt
        if world == 1 and not args.no_dropin and not args.with_color and args.precision == "f32":
            out["dropin"] = dropin_surface(device, args.steps, args.warmup, resolutions, args.passes)
            out["latency_ms_single_frame"] = out["dropin"]["latency_ms_single_frame"]
            if args.soak > 0:
                out["dropin"]["soak"] = soak(device, args.soak, resolutions)
        if world == 1 and not args.no_cpu_baseline and not args.with_color and args.levels == 5:
            # bounded thread count: torch-CPU convs at batch 1 collapse when oversubscribed
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 32))
            out["cpu_baseline_reference_ops"] = out["cpu_baseline"].pop("reference_ops")
            out["cpu_baseline_reference"] = CPU_BASELINE_REFERENCE
        launch_log = os.environ.get("MONOPORT_BENCH_LAUNCH_LOG")
        if launch_log:  # tools/profile_summary.py: per-launch (ms, points) of the roofline leg
            with open(launch_log, "w") as f:
                json.dump({"launch_ms": [float(v) for v in roof["launch_ms"]],
                           "launch_points": [int(v) for v in roof["launch_pts"]],
                           "levels": len(resolutions), "frames_per_launch": min(batch, MAX_RECON_BATCH),
                           "flop_per_point": FLOP_PER_POINT_SKIP_TABLE if skip_on else FLOP_PER_POINT}, f)
        print(json.dumps(out), flush=True)
    pipe.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
