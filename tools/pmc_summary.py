"""Print per-dispatch counter values of the fused query kernels from a rocprofv3 --pmc output dir."""
import csv, glob, os, sys
from collections import defaultdict
rows = defaultdict(dict)
grid = {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if "pifu_query" in r["Kernel_Name"] or "skip_table_kernel" in r["Kernel_Name"]:
                d = int(r["Dispatch_Id"])
                rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                grid[d] = r["Grid_Size"]
for d in sorted(rows):
    print(d, "grid", grid[d], " ".join("%s=%.4g" % kv for kv in sorted(rows[d].items())))
