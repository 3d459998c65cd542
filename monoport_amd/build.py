"""Build libmonoport_hip.so (hipcc, gfx950 only) in-tree: ``python -m monoport_amd.build``.

The shared library lands in monoport_amd/lib/ so that it travels with the source snapshot to
the GPU box; nothing is JIT-compiled at import time.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libmonoport_hip.so")
SOURCES = ["api.hip", "pack.hip", "query.hip", "query_small.hip", "query_table.hip", "query16.hip", "octree.hip", "vertices.hip", "mcubes.hip",
           "encoder_ops.hip", "conv3x3.hip", "conv_wino.hip", "convim2col.hip", "plan.hip", "clock_probe.hip"]
HEADERS = [os.path.join(CSRC, "mp_internal.h"), os.path.join(CSRC, "query_common.h"),
           os.path.join(CSRC, "encoder_kernels.h"), os.path.join(CSRC, "gn_tail.h"),
           os.path.join(CSRC, "query_mfma.h"),
           os.path.join(os.path.dirname(HERE), "include", "monoport_hip.h")]
# -ffp-contract=off: parity with the reference is op-order parity; hipcc's default (fast) lets the
# backend fuse any a*b+c, pragmas notwithstanding.  FMAs are requested explicitly (fmaf, MFMA).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        extra = [os.path.join(CSRC, "mc_tables.h")] if s == "mcubes.hip" else []
        if force or _stale(obj, [src] + HEADERS + [e for e in extra if os.path.exists(e)]):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
