"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python bench.py ...`
output directory into the two small files committed under profiles/:

  <prefix>_kernel_stats.csv   top kernels by total time (rocprofv3's own kernel_stats.csv, truncated)
  <prefix>_query_launches.txt every fused-query launch in time order with its duration, plus the
                              mean duration per octree level over the single-stream roofline leg
                              (the launches bench.py brackets with HIP events)

    python tools/profile_summary.py DIR profiles/r01f_bench "command line that was profiled" [N_LEG [LAUNCH_LOG]]

N_LEG = launches of the roofline leg (levels x batches); LAUNCH_LOG = the JSON bench.py writes when
MONOPORT_BENCH_LAUNCH_LOG is set: per launch of that leg the HIP-event duration and the POINT COUNT,
so that the per-launch TFLOP/s can be recomputed from the file alone.
"""
import csv
import glob
import json
import os
import sys

FLOP_PER_POINT = 2363906


def main(directory, prefix, command):
    stats = glob.glob(os.path.join(directory, "**", "*kernel_stats.csv"), recursive=True)[0]
    with open(stats) as f:
        lines = f.readlines()
    with open(prefix + "_kernel_stats.csv", "w") as f:
        f.writelines(lines[:41])
    trace = glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    with open(trace) as f:
        for r in csv.DictReader(f):
            # the f32 kernels: 32-point tiles (netG) / 64-point tiles -- not pifu_query16_kernel
            if ("pifu_query_tab" in r["Kernel_Name"] or "pifu_query_t32_kernel<1>" in r["Kernel_Name"]
                    or "pifu_query_kernel<256, 1" in r["Kernel_Name"]):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40],
                             r["Grid_Size_X"], r["Workgroup_Size_X"]))
    rows.sort()
    # An octree level whose point counts live on the device is launched on BOTH tile sizes; the
    # kernel the 2048-tile gate excludes leaves at its first instruction (a few us).  Those empty
    # dispatches are not query launches: drop them from the list (counted in the header).
    n_all = len(rows)
    rows = [r for r in rows if (r[1] - r[0]) / 1e3 >= 25.0]
    durs = [(e - s) / 1e3 for s, e, *_ in rows]
    with open(prefix + "_query_launches.txt", "w") as f:
        f.write("rocprofv3 --kernel-trace --stats --output-format csv -- %s\n" % command)
        f.write("%d fused-query launches (+ %d gate-excluded empty dispatches of < 25 us, not listed); "
                "duration (us), grid, kernel -- in start order\n" % (len(rows), n_all - len(rows)))
        for (s, e, name, grid, wg), d in zip(rows, durs):
            f.write("%10.1f  grid %-8s %s\n" % (d, grid, name))
        # the roofline leg (ONE stream, every launch bracketed by HIP events) is found by its
        # durations: the window of N_LEG consecutive launches that matches bench.py's HIP-event
        # log best (the timed passes overlap three streams, so their launches run longer)
        if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
            log = json.load(open(sys.argv[5]))
            ev = [ms * 1e3 for ms in log["launch_ms"]]
            n_leg = len(ev)
            best, at = None, 0
            for i in range(0, len(durs) - n_leg + 1):
                err = sum(abs(durs[i + k] - ev[k]) / ev[k] for k in range(n_leg))
                if best is None or err < best:
                    best, at = err, i
            leg = durs[at:at + n_leg]
            f.write("\nroofline leg = launches %d..%d (1-based, of the list above; mean relative difference to "
                    "bench.py's HIP-event durations %.2f %%): mean %.1f us\n"
                    % (at + 1, at + n_leg, 100 * best / n_leg, sum(leg) / len(leg)))
            fpp = int(log.get("flop_per_point", FLOP_PER_POINT))  # executed per point (skip tables: fewer)
            f.write("per launch: level, points (all frames of the launch), rocprofv3 duration, HIP-event "
                    "duration, TFLOP/s = points x %d EXECUTED FLOP / rocprofv3 duration%s\n"
                    % (fpp, "" if fpp == FLOP_PER_POINT else
                       " (the reference's MLP has %d per point: x %.3f for the algorithmic rate)"
                       % (FLOP_PER_POINT, FLOP_PER_POINT / fpp)))
            tot_f = tot_t = 0.0
            for i, (d, ms, pts) in enumerate(zip(leg, log["launch_ms"], log["launch_points"])):
                tf = pts * fpp / (d * 1e-6) / 1e12
                tot_f += pts * fpp
                tot_t += d * 1e-6
                f.write("  level %d  %9d points  %9.1f us  (events %9.1f us)  %6.1f TFLOP/s  frac %.3f\n"
                        % (i % log["levels"], pts, d, ms * 1e3, tf, tf / 157.3))
            f.write("  leg total: %.1f TFLOP/s = %.3f of the 157.3 TFLOP/s f32 MFMA peak\n"
                    % (tot_f / tot_t / 1e12, tot_f / tot_t / 1e12 / 157.3))
            print("roofline leg mean us:", sum(leg) / len(leg))
    print("launches:", len(rows), "mean us:", sum(durs) / max(1, len(durs)))


if __name__ == "__main__":
    main(*sys.argv[1:4])
