"""Per octree level: points, time and fraction of the f32 MFMA roof of the fused-query launches of bench.py's
roofline leg, from the log bench.py writes when MONOPORT_BENCH_LAUNCH_LOG is set.

    MONOPORT_BENCH_LAUNCH_LOG=log.json python bench.py --no-extras --no-cpu-baseline; python tools/launch_levels.py log.json
"""
import json
import sys

import numpy as np

PEAK = 157.3e12


def main():
    d = json.load(open(sys.argv[1]))
    L = d["levels"]
    ms = np.array(d["launch_ms"]).reshape(-1, L)
    pts = np.array(d["launch_points"], dtype=np.float64).reshape(-1, L)
    fpp = d["flop_per_point"]
    print("%d launches per level, %d frames per launch, %d FLOP per point" % (ms.shape[0], d["frames_per_launch"], fpp))
    for l in range(L):
        frac = pts[:, l] * fpp / (ms[:, l] * 1e-3) / PEAK
        print("level %d: %8.0f points per launch, %7.3f ms, frac %.3f (%.3f .. %.3f), %4.1f %% of the time"
              % (l, pts[:, l].mean(), ms[:, l].mean(), pts[:, l].sum() * fpp / (ms[:, l].sum() * 1e-3) / PEAK,
                 frac.min(), frac.max(), 100 * ms[:, l].sum() / ms.sum()))
    print("all launches: frac %.4f" % (pts.sum() * fpp / (ms.sum() * 1e-3) / PEAK))


if __name__ == "__main__":
    main()
