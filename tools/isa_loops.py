"""List the loops of a kernel in hipcc -S output with their MFMA / memory / spill instruction counts.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/k.s FILE.hip
    python tools/isa_loops.py /tmp/k.s KERNEL_NAME_SUBSTRING
"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and sys.argv[2] in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
print(body[0][:90], "-- %d lines" % len(body))
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = body[labels[m.group(1)]:i]
        cnt = lambda pat: sum(bool(re.search(pat, x)) for x in seg)
        if cnt("v_mfma"):
            print("  loop %5d-%5d: mfma %4d  global_load %3d  ds_read %3d  ds_write %3d  scratch_load %3d  "
                  "scratch_store %3d  v_accvgpr %3d  s_barrier %d"
                  % (labels[m.group(1)], i, cnt("v_mfma"), cnt("global_load"), cnt(r"ds_read|ds_load"),
                     cnt(r"ds_write|ds_store"), cnt("scratch_load"), cnt("scratch_store"),
                     cnt("v_accvgpr"), cnt("s_barrier")))
