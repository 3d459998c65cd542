"""Workload for the rocprofv3 --pmc passes behind bench.py's roofline.traffic.

    run   : one pipeline slot, BATCH frames per batch, 2 timed-style batches (after warm-up).
            A batch is 9 fused-query dispatches: level 0 (host-side counts) on the kernel the launcher
            picked, levels 1-4 (device-side counts) on BOTH tile sizes, of which the one the gate
            excludes leaves at once and moves no bytes -- a level's traffic is the sum of its pair
    parse : counter_collection.csv of the FETCH_SIZE and WRITE_SIZE passes -> profiles/*.json

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python tools/traffic_probe.py run
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python tools/traffic_probe.py run
  python tools/traffic_probe.py parse gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r03_query_traffic.json
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# frames per slot submission / launch: MONOPORT_TRAFFIC_BATCH (bench.py: 32 by default, 20 at the driver's --steps 20)
BATCH = int(os.environ.get("MONOPORT_TRAFFIC_BATCH", "16"))
# MONOPORT_TRAFFIC_LEVELS=6 MONOPORT_TRAFFIC_PRECISION=f16w: BASELINE configs[4] (513^3, fp16 weights: pifu_query16_kernel)
LEVELS = int(os.environ.get("MONOPORT_TRAFFIC_LEVELS", "5"))
PRECISION = os.environ.get("MONOPORT_TRAFFIC_PRECISION", "f32")


def run():
    import torch
    import bench
    from monoport_amd import synthetic as syn
    from monoport_amd.recon import pifu_calib
    dev = torch.device("cuda", 0)
    res = bench.RESOLUTIONS + ([513] if LEVELS == 6 else [])
    pipe = bench.make_pipeline(dev, 1, False, res, False, PRECISION, BATCH)
    images = [torch.from_numpy(syn.synthetic_image(s))[None].to(dev) for s in range(BATCH)]
    for k in range(2):
        calibs = [pifu_calib(*syn.scene_camera(3 * (BATCH * k + b)), device=dev) for b in range(BATCH)]
        slot = pipe.submit(images, calibs)
        slot.wait()
    print("points per level (last batch):", slot.status[:, 1:].sum(0).tolist())


def counter_rows(directory, counter):
    """Per-level counter values of the last 2 batches: [level 0, ..., level 4] x 2."""
    rows = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                if r["Counter_Name"] == counter and ("pifu_query_tab" in name or "pifu_query_t32_kernel" in name
                                                     or "pifu_query_kernel" in name or "pifu_query16" in name):
                    rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                                 "t32" in name or "tab" in name or "pifu_query16" in name))
    rows.sort()
    # with the skip tables (default) every level is ONE dispatch of the table kernel per chunk of <= 32 frames
    # (kMaxFrames; the chunks of a level are added up); on the plain path
    # (MONOPORT_SKIP_TABLE=off) levels 1-4 are a gated pair
    chunks = (BATCH + 31) // 32
    if not any(not t32 for _, _, t32 in rows[-2 * LEVELS * chunks:]):
        rows = rows[-2 * LEVELS * chunks:]
        assert len(rows) == 2 * LEVELS * chunks, len(rows)
        vals = [v for _, v, _ in rows]
        # dispatch order inside a batch: level 0 of every chunk, then level 1 of every chunk, ... (mp_recon_batch
        # is called per chunk: chunk-major) -- pipeline.py calls mp_recon_batch once per chunk, so it is chunk-major
        out = []
        for b in range(2):
            per = vals[b * LEVELS * chunks:(b + 1) * LEVELS * chunks]
            out += [sum(per[c * LEVELS + l] for c in range(chunks)) for l in range(LEVELS)]
        return out
    per_batch = 1 + 2 * (LEVELS - 1)
    rows = rows[-2 * per_batch:]
    assert len(rows) == 2 * per_batch, len(rows)
    out = []
    for b in range(2):
        r = rows[b * per_batch:(b + 1) * per_batch]
        out.append(r[0][1])
        for l in range(1, LEVELS):
            pair = r[1 + 2 * (l - 1):3 + 2 * (l - 1)]
            assert pair[0][2] != pair[1][2], "a device-count level is one dispatch of each kernel"
            out.append(pair[0][1] + pair[1][1])
    return out


def parse_counter(directory, counter, out_path):
    """Any other per-launch counter of the same workload (e.g. TCP_TCC_READ_REQ_sum: the requests the L1s send to
    the L2s -- the weight stream of the f16 kernels): per-level mean of the last two batches, raw counts."""
    vals = counter_rows(directory, counter)
    per_level = [(vals[l] + vals[LEVELS + l]) / 2 for l in range(LEVELS)]
    out = {"counter": counter, "slot_batch": BATCH, "levels": LEVELS, "precision": PRECISION, "per_level": per_level}
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


def parse(fetch_dir, write_dir, out_path):
    fetch = counter_rows(fetch_dir, "FETCH_SIZE")
    write = counter_rows(write_dir, "WRITE_SIZE")
    per_level_fetch = [(fetch[l] + fetch[LEVELS + l]) / 2 for l in range(LEVELS)]
    per_level_write = [(write[l] + write[LEVELS + l]) / 2 for l in range(LEVELS)]
    # rocprofv3 reports both in KB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: wide
    # coalesced reads are tallied at half their bytes on gfx950), WRITE_SIZE taken as is
    bytes_level = [1024.0 * (2 * f + w) for f, w in zip(per_level_fetch, per_level_write)]
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of "
                  "tools/traffic_probe.py run: one slot, %d frames per mp_recon_batch, mean of the "
                  "last 2 batches, one fused-query launch per octree level (skip-table kernel by default; on "
                  "the plain path 32-point tiles below 2048 64-point tiles per launch, 64-point tiles above)" % BATCH,
        "unit": "KB (rocprofv3 counter units, x1024 bytes)",
        "FETCH_SIZE_per_level": per_level_fetch,
        "WRITE_SIZE_per_level": per_level_write,
        "correction": "FETCH_SIZE doubled (128-B requests tallied at 64 B for 16 B/lane coalesced "
                      "reads on gfx950), WRITE_SIZE as is",
        "slot_batch": BATCH, "levels": LEVELS, "precision": PRECISION,
        "launches_per_level": (BATCH + 31) // 32,
        "bytes_per_launch_avg": sum(bytes_level) / (LEVELS * ((BATCH + 31) // 32)),
        "bytes_per_level_launch": bytes_level,
    }
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "counter":
        parse_counter(*sys.argv[2:5])
    else:
        parse(*sys.argv[2:5])
