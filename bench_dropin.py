"""The drop-in legs of bench.py (split out in round 6): the reference's own call surface -- the processors list of
RTL/main.py:326-452 on the thread-per-stage pipeline -- measured for throughput (per-frame stages, coalescing
stages), single-frame latency and, new, as a SOAK: the per-frame pipeline run for a fixed time on rotating inputs
with the latency distribution and the memory the process holds per 10-second window."""
import os
import sys
import time

import numpy as np
import torch

from bench_common import B_MAX, B_MIN, N_IMAGES, build_netg
from monoport_amd import ops, synthetic as syn
from monoport_amd.recon import pifu_calib


def dropin_surface(device, n_frames, n_warm, resolutions, passes=5, legs=("per_frame", "per_frame_trusted", "coalesced")):
    """The reference's own call surface, as RTL/main.py:326-452 drives it: the processors=[...]
    list (H2D, camera, pifu_calib, input normalisation, netG.filter, reconEngine =
    Seg3dLossless(query_func) with its per-frame host sync, forward_vertices with its .item(),
    colorization) on the thread-per-stage pipeline, eager encoder.  Two modes, `passes` runs each
    (median / min / max): `per_frame_stages` = one frame per stage call, exactly the reference's
    structure; and the headline of this leg, the same list with the three heavy stages wrapped in
    stage_pipeline.Coalesced -- when frames queue up in front of a stage it serves up to 8 of them in
    one call (a batched netG.filter, Seg3dLossless.forward_many = one mp_recon_batch, one host sync
    for all vertex counts); per-frame results are unchanged.  Also the latency of a single frame run
    stage by stage."""
    from monoport_amd.implicit_seg.functional import Seg3dLossless
    from monoport_amd.recon import colorization, forward_vertices, forward_vertices_many
    from monoport_amd.stage_pipeline import Coalesced, StagePipeline
    netG, _ = build_netg(device)
    planes = torch.from_numpy(syn.body_feature_planes(128, 128)).to(device)

    def query_func(points, im_feat_list, calib_tensor):  # RTL/main.py:169-183
        assert len(points) == 1
        samples = points.repeat(1, 1, 1)
        samples = samples.permute(0, 2, 1)
        return netG.query(im_feat_list, points=samples, calibs=calib_tensor)[0]

    # two engines: the class default validates query_func on EVERY frame (one extra 17^3 query + host sync;
    # what a maintainer gets by swapping the import); validate="first" trusts a closure after three agreeing
    # frames (re-checked every 32nd) and is what lets a coalescing stage batch frames (forward_many)
    engines = {v: Seg3dLossless(query_func=query_func, b_min=np.array([B_MIN], np.float32),
                                b_max=np.array([B_MAX], np.float32), resolutions=resolutions,
                                balance_value=0.5, use_cuda_impl=False, faster=True, validate=v).to(device)
               for v in ("always", "first")}
    mean, std = 0.5, 0.5
    r_last = resolutions[-1]

    def filt(d):
        feats = netG.filter(d["input_netG"])
        feats[-1][0][0, 0:2].copy_(planes)  # synthetic body planes, as in the headline run
        return {**d, "feat_tensor_G": feats}

    debug = os.environ.get("MONOPORT_DROPIN_DEBUG") == "1"
    call_log = []

    def logged(name, fn):
        if not debug:
            return fn

        def wrapper(x):
            t0 = time.perf_counter()
            out = fn(x)
            call_log.append((name, len(x) if isinstance(x, list) else 1, time.perf_counter() - t0))
            return out
        return wrapper

    def filt_many(ds):
        feats = netG.filter(torch.cat([d["input_netG"] for d in ds]))
        out = []
        for i, d in enumerate(ds):
            fi = [[f[i:i + 1] for f in stage] for stage in feats]
            fi[-1][0][0, 0:2].copy_(planes)
            out.append({**d, "feat_tensor_G": fi})
        return out

    def recon_many(ds):
        sdfs = engines["first"].forward_many([dict(im_feat_list=d["feat_tensor_G"], calib_tensor=d["calib_tensor"]) for d in ds])
        return [{**d, "sdf": sdf} for d, sdf in zip(ds, sdfs)]

    def vertices_many(ds):
        vs = forward_vertices_many([d["sdf"] for d in ds], direction="front")
        return [{**d, **dict(zip(["X", "Y", "Z", "norm"], v))} for d, v in zip(ds, vs)]

    def processors(step, coalesce=False, validate="always"):
        engine = engines["first" if coalesce else validate]

        def camera(d):
            ext, intr = syn.scene_camera(3 * step[0])
            step[0] += 1
            return {**d, "extrinsic": ext, "intrinsic": intr}
        def recon_one(d):
            return {**d, "sdf": engine(im_feat_list=d["feat_tensor_G"], calib_tensor=d["calib_tensor"])}

        def vertices_one(d):
            return {**d, **dict(zip(["X", "Y", "Z", "norm"], forward_vertices(d["sdf"], direction="front")))}

        wrap = ((lambda one, many, name: Coalesced(logged(name, one), logged(name, many), max_batch=CO_BATCH, max_pending=CO_PENDING)) if coalesce
                else (lambda one, many, name: logged(name, one)))
        return [
            lambda data: {"input": data.to(device, non_blocking=True)},                    # main.py:327
            camera,                                                                       # :330-336
            lambda d: {**d, "calib_tensor": pifu_calib(d["extrinsic"], d["intrinsic"], device=device)},
            lambda d: {**d, "input_netG": (((d["input"][:, 0:3] * 0.5 + 0.5) - mean) / std)
                       * d["input"][:, 3:4]},                                             # :353-357
            wrap(filt, filt_many, "filter"),                                              # :367-370
            wrap(recon_one, recon_many, "recon"),                                         # :390-395
            wrap(vertices_one, vertices_many, "vertices"),                                # :401-406
            lambda d: {**d, "render_norm": colorization(None, None, d["X"], d["Y"], d["Z"],
                                                        d["calib_tensor"], d["norm"],
                                                        resolution=r_last)},              # :418-428
        ]

    # coalescing stages: frames served per call at most / frames in flight (MONOPORT_DROPIN_COALESCE="batch,in_flight")
    # + batches of a stage in flight on the GPU at once (Coalesced(max_pending=...), 0 = unthrottled)
    CO_BATCH, CO_IN_FLIGHT, CO_PENDING = (int(v) for v in (os.environ.get("MONOPORT_DROPIN_COALESCE", "16,48,1") + ",1").split(",")[:3])
    frames = []
    for i in range(N_IMAGES):
        img = torch.from_numpy(syn.synthetic_image(i))
        mask = (img.abs().sum(0, keepdim=True) > 0).float()
        frames.append(torch.cat([img, mask], 0)[None].pin_memory())

    # single-frame latency: one frame through the stages, one after the other, nothing else on
    # the GPU; median of 7 after a warm-up
    procs = processors([0])
    lat = []
    with torch.no_grad():
        for i in range(3 + 7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d = frames[i % N_IMAGES]
            for p in procs:
                d = p(d)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
    assert d["render_norm"] is not None
    latency_ms = float(np.median(lat[3:])) * 1e3

    # throughput: the same list on the stage pipeline (thread + stream per stage, FIFO order)
    n_frames = max(n_frames, 96)  # long against the pipeline's fill and drain, which are INSIDE the timed region
    passes = max(passes, 5)

    def one_pass(coalesce, in_flight, validate):
        engine = engines["first" if coalesce else validate]

        def source():
            for i in range(n_frames):
                yield frames[i % N_IMAGES]

        # the clock runs from an EMPTY pipeline to an empty pipeline (fill and drain included): starting it
        # after a few warm-up outputs would count frames that are already half way through the stages
        out_count, last = 0, None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for d in StagePipeline(source(), processors([0], coalesce, validate), device=device, max_in_flight=in_flight):
                out_count += 1
                last = d
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        assert out_count == n_frames and engine.last_path == "fused" and last["render_norm"] is not None
        if debug:
            st = torch.cuda.memory_stats()
            print("dropin pass coalesce=%s: %.1f recon/s; device allocs %d frees %d retries %d, reserved %.1f GB; %d calls, "
                  "batch sizes %s; slow calls: %s"
                  % (coalesce, n_frames / elapsed, st.get("num_device_alloc", -1), st.get("num_device_free", -1),
                     st.get("num_alloc_retries", -1), torch.cuda.memory_reserved() / 2 ** 30, len(call_log),
                     sorted(set(b for _, b, _ in call_log)),
                     " ".join("%s x%d %.0fms" % (n, b, 1e3 * t) for n, b, t in call_log if t > 0.1)), file=sys.stderr, flush=True)
            call_log.clear()
        return elapsed

    def mode(coalesce, in_flight, validate="first"):
        if coalesce:
            # untimed: first use of every encoder batch size the coalescing filter stage can meet, ON THAT STAGE'S
            # STREAM (stage 4 of the list) -- torch's allocator pools blocks per stream, and a first batched encoder
            # pass on a cold pool costs 1.5-1.8 s of hipMalloc (profiles/r04g_dropin_passes.txt)
            from monoport_amd.stage_pipeline import stage_stream
            with torch.no_grad(), torch.cuda.stream(stage_stream(device, 4)):
                for b in range(1, CO_BATCH + 1):
                    netG.filter(torch.zeros((b, 3, 512, 512), device=device))
            torch.cuda.synchronize()
        one_pass(coalesce, in_flight, validate)  # untimed
        runs = sorted(one_pass(coalesce, in_flight, validate) for _ in range(passes))
        med = runs[len(runs) // 2]
        return {"value": n_frames / med, "unit": "recon/s", "ms_per_step": med / n_frames * 1e3,
                "passes": {"n": passes, "value_min": n_frames / runs[-1], "value_median": n_frames / med,
                           "value_max": n_frames / runs[0]},
                "frames_in_flight": in_flight, "validate": "first" if coalesce else validate}

    per_frame = mode(False, 8, "always")
    if "coalesced" not in legs:  # probes (tools/per_frame_overlap_probe.py): the per-frame leg and the latency only
        return {"per_frame_stages": per_frame, "latency_ms_single_frame": latency_ms, "frames": n_frames}
    per_frame_trusted = mode(False, 8, "first")
    co = mode(True, CO_IN_FLIGHT)
    return {
        "surface": "RTL/main.py processors list on StagePipeline: Seg3dLossless(query_func) + forward_vertices + "
                   "colorization, eager encoder; netG.filter / reconEngine / forward_vertices as Coalesced stages "
                   "(up to %d queued frames per call), %d frames in flight" % (CO_BATCH, CO_IN_FLIGHT),
        **co,
        "per_frame_stages": {**per_frame,
                             "surface": "the same list, one frame per stage call (the reference's structure), batch 1, "
                                        "8 frames in flight; Seg3dLossless as constructed by RTL/main.py:188-195 (class "
                                        "default validate='always': query_func checked on every frame)"},
        "per_frame_stages_trusted": {**per_frame_trusted,
                                     "surface": "the same with Seg3dLossless(..., validate='first')"},
        "latency_validate": "always",
        "latency_ms_single_frame": latency_ms,
        "latency_ms_min": float(np.min(lat[3:])) * 1e3,
        "frames": n_frames,
    }


def _percentiles(values_ms):
    v = np.asarray(values_ms, np.float64)
    return {"p50": float(np.percentile(v, 50)), "p90": float(np.percentile(v, 90)), "p99": float(np.percentile(v, 99)),
            "max": float(v.max()), "mean": float(v.mean()), "n": int(v.size)}


def soak(device, seconds, resolutions, window_s=None, n_inputs=64, in_flight=8, empty_every=37, raise_at=None,
         latency_sweep=(1, 2, 4), sweep_seconds=1.5):
    """The reference's steady-state operating mode -- an endless ``for data_dict in loader`` at one frame per stage
    call (RTL/main.py:487, RTL/dataloader.py:1026-1053) -- run for ``seconds``: the processors list of
    RTL/main.py:326-452 on the per-frame StagePipeline (Seg3dLossless as the reference constructs it, class default
    validate='always'), fed ``n_inputs`` distinct images and cameras in rotation, every ``empty_every``-th frame an
    EMPTY scene (the engine returns None and the None travels through forward_vertices / colorization, RTL/recon.py:
    32-33, RTL/main.py:214-215).  The consumer copies every render to the host (the reference displays it).

    Reported: frames, recon/s and the latency distribution (admission to the pipeline -> render on the host; with
    ``in_flight`` frames admitted at once this includes the time a frame queues behind its predecessors) overall and
    per ``window_s`` window, and per window what the process holds: torch's reserved / allocated bytes, the C side's
    scratch arenas / packed weights / arenas / registered skip tables (mp_memory_stats) and the live encoder plans.
    ``flat``: the last window's figures equal the second window's (the first one warms up).  ``window_s`` defaults
    to a sixth of the run, at least 2.5 s.
    ``latency_sweep``: after the soak, the same pipeline for ``sweep_seconds`` each at these numbers of frames in
    flight -- by Little's law the admission-to-render latency at k frames in flight is k / throughput, so the
    8-in-flight figure of the soak is a queueing time; the sweep shows what a caller gets who admits fewer.
    ``raise_at``: (test hook) make the recon stage raise on that frame: the error must reach the consumer."""
    from monoport_amd.implicit_seg.functional import Seg3dLossless
    from monoport_amd.recon import colorization, forward_vertices
    netG, _ = build_netg(device)
    planes = torch.from_numpy(syn.body_feature_planes(128, 128)).to(device)
    empty_planes = torch.empty_like(planes)
    empty_planes[0], empty_planes[1] = -4.0, 4.0  # front behind back everywhere: nothing is inside
    r_last = resolutions[-1]

    def query_func(points, im_feat_list, calib_tensor):  # RTL/main.py:169-183
        assert len(points) == 1
        samples = points.repeat(1, 1, 1)
        samples = samples.permute(0, 2, 1)
        return netG.query(im_feat_list, points=samples, calibs=calib_tensor)[0]

    engine = Seg3dLossless(query_func=query_func, b_min=np.array([B_MIN], np.float32),
                           b_max=np.array([B_MAX], np.float32), resolutions=resolutions, balance_value=0.5,
                           use_cuda_impl=False, faster=True).to(device)
    frames = []
    for i in range(n_inputs):
        img = torch.from_numpy(syn.synthetic_image(i))
        mask = (img.abs().sum(0, keepdim=True) > 0).float()
        frames.append(torch.cat([img, mask], 0)[None].pin_memory())
    cameras = [syn.scene_camera(5 * i) for i in range(n_inputs)]

    def filt(d):
        feats = netG.filter(d["input_netG"])
        feats[-1][0][0, 0:2].copy_(empty_planes if d["empty"] else planes)
        return {**d, "feat_tensor_G": feats}

    procs = [
        lambda item: {"index": item[0], "empty": item[2], "camera": item[3],
                      "input": item[1].to(device, non_blocking=True)},                          # main.py:327
        lambda d: {**d, "extrinsic": d["camera"][0], "intrinsic": d["camera"][1]},              # :330-336
        lambda d: {**d, "calib_tensor": pifu_calib(d["extrinsic"], d["intrinsic"], device=device)},
        lambda d: {**d, "input_netG": (((d["input"][:, 0:3] * 0.5 + 0.5) - 0.5) / 0.5) * d["input"][:, 3:4]},
        filt,                                                                                   # :367-370
        None,                                                                                   # :390-395 (procs_for)
        lambda d: {**d, **dict(zip(["X", "Y", "Z", "norm"], forward_vertices(d["sdf"], direction="front")))},
        lambda d: {**d, "render_norm": colorization(None, None, d["X"], d["Y"], d["Z"], d["calib_tensor"],
                                                    d["norm"], resolution=r_last)},              # :418-428
    ]
    if window_s is None:
        window_s = max(2.5, float(seconds) / 6.0)

    def run_once(seconds, in_flight, window_s, raise_at):
        return _soak_run(device, procs_for(raise_at), frames, cameras, n_inputs, empty_every, seconds, in_flight, window_s,
                         held)

    def procs_for(raise_at_frame):
        def recon_stage(d):
            if raise_at_frame is not None and d["index"] == raise_at_frame:
                raise ValueError("injected failure at frame %d" % raise_at_frame)
            return {**d, "sdf": engine(im_feat_list=d["feat_tensor_G"], calib_tensor=d["calib_tensor"])}
        return procs[:5] + [recon_stage] + procs[6:]

    def held():
        st = ops.memory_stats(device)
        return {"torch_reserved": int(torch.cuda.memory_reserved(device)), "torch_allocated": int(torch.cuda.memory_allocated(device)),
                "mp_arena_bytes": st["arena_bytes"], "mp_weight_bytes": st["weight_bytes"], "mp_arenas": st["arenas"],
                "mp_skip_tables": st["skip_tables"], "encoder_plans": netG.image_filter.plan_count()}

    res = run_once(seconds, in_flight, window_s, raise_at)
    res["surface"] = ("RTL/main.py processors list on the per-frame StagePipeline (validate='always'), %d distinct images / "
                      "cameras in rotation, every %dth frame an empty scene, %d frames in flight, renders copied to the host"
                      % (n_inputs, empty_every, in_flight))
    if raise_at is None and latency_sweep:
        res["latency_by_frames_in_flight"] = {}
        for k in latency_sweep:
            r = run_once(sweep_seconds, k, sweep_seconds, None)
            res["latency_by_frames_in_flight"][str(k)] = {"value": r["value"], "latency_ms": r["latency_ms"]}
    return res


def _soak_run(device, procs, frames, cameras, n_inputs, empty_every, seconds, in_flight, window_s, held):
    from monoport_amd.stage_pipeline import StagePipeline
    admitted = {}
    t_end = [None]

    def source():
        i = 0
        while time.perf_counter() < t_end[0]:
            yield (i, frames[i % n_inputs], empty_every > 0 and i % empty_every == empty_every - 1, cameras[i % n_inputs])
            i += 1

    def admit(item):  # stage 0 (the H2D copy, RTL/main.py:327): the frame has a slot and enters the pipeline NOW --
        admitted[item[0]] = time.perf_counter()  # the feeder pulls the source one item ahead of the in-flight semaphore
        return first_stage(item)

    first_stage = procs[0]
    procs = [admit] + list(procs[1:])
    tables0 = held()["mp_skip_tables"]  # what the process registered before this run (other legs' pipelines)
    windows, lat_all, lat_win, lat_steady, none_count = [], [], [], [], 0
    pipe = StagePipeline(source(), procs, device=device, max_in_flight=in_flight)
    error = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_end[0] = t0 + float(seconds)
    t_win, n_win = t0, 0
    try:
        with torch.no_grad():
            for d in pipe:
                if d["render_norm"] is None:
                    none_count += 1
                    assert d["empty"], "a non-empty scene came back as None"
                else:
                    assert not d["empty"]
                    d["render_norm"].cpu()  # the consumer's D2H (the reference shows the image)
                now = time.perf_counter()
                lat = (now - admitted.pop(d["index"])) * 1e3
                lat_all.append(lat)
                lat_win.append(lat)
                if windows:  # past the first window: plans recorded, allocator pools and arenas grown
                    lat_steady.append(lat)
                n_win += 1
                if now - t_win >= window_s:
                    windows.append({"t_s": now - t0, "frames": n_win, "value": n_win / (now - t_win),
                                    "latency_ms": _percentiles(lat_win), **held()})
                    t_win, n_win, lat_win = now, 0, []
    except RuntimeError as e:  # a StageError re-raised in the consumer (stage_pipeline.StageError.reraise)
        error = str(e)
    elapsed = time.perf_counter() - t0
    for t in pipe._threads:
        t.join(timeout=10)
    alive = sum(t.is_alive() for t in pipe._threads)
    # what must not move after the first window: bytes held and the things that own bytes.  The registered skip tables
    # are the frames in flight between the encoder and the vertex stage (a table lives as long as its frame's feature
    # map): their number breathes with the pipeline's occupancy but is bounded by the frames in flight.
    # torch's caching allocator may still add a segment after the first window (which stream frees a block first is a
    # matter of timing): its reserved bytes may grow by <= 10 % over the rest of the run, the C side's not at all.
    keys = ("mp_arena_bytes", "mp_weight_bytes", "mp_arenas", "encoder_plans")
    flat = (len(windows) >= 3 and all(windows[-1][k] == windows[1][k] for k in keys)
            and windows[-1]["torch_reserved"] <= 1.10 * windows[1]["torch_reserved"]
            and all(w["mp_skip_tables"] <= tables0 + in_flight + 2 for w in windows[1:]))
    return {
        "seconds": elapsed, "frames": len(lat_all), "none_frames": none_count,
        "value": len(lat_all) / elapsed, "unit": "recon/s",
        "latency_ms": _percentiles(lat_all) if lat_all else None,
        "latency_ms_after_first_window": _percentiles(lat_steady) if lat_steady else None,
        "latency_definition": "admission to the pipeline -> render on the host, %d frames in flight" % in_flight,
        "windows": windows, "window_s": window_s,
        "flat_after_warmup": bool(flat), "flat_keys": list(keys) + ["torch_reserved within 10 %","mp_skip_tables <= those registered before the run + frames in flight + 2"],
        "error": error, "stage_threads_alive_after": int(alive),
    }
