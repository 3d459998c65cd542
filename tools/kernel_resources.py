"""VGPR / spill / scratch / occupancy of every kernel in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
   python tools/kernel_resources.py monoport_amd/csrc/conv3x3.hip [filter]"""
import os, re, subprocess, sys
src = os.path.abspath(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[3:]
err = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
cur = None; rows = {}
for line in err.splitlines():
    m = re.search(r"remark: ([A-Za-z \[\]/]+): (.*?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur: rows[cur][k] = v
for name, r in rows.items():
    if flt and flt not in name: continue
    print("%-70s VGPR %3s AGPR %3s spill %3s scratch %4s occ %s sgpr-spill %s" % (
        name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
        r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill")))
