#!/bin/bash
# per-launch timeline of ONE batch-1 netG encoder pass (the call a per-frame filter stage makes): rocprofv3 kernel trace of
# tools/enc_profile.py 1, the launches of the last pass in start order with duration, grid and the gap to the previous launch
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r06i}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
MONOPORT_ENCODER_BRANCHES=${BR:-on} rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python $R/tools/enc_profile.py 1 > $out/enc_prof.log 2>&1
cd $R
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $out/encoder_b1_timeline_${BR:-on}.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows)//12
last=rows[-n:]
t0=int(last[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in last)
dur=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in last)
print("batch-1 encoder pass: %d launches, wall %.3f ms, sum of kernel durations %.3f ms" % (n,(t1-t0)/1e6,dur/1e6))
by=collections.defaultdict(lambda:[0,0])
prev_end=t0
for r in last:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    g=int(r["Grid_Size_X"])//max(int(r["Workgroup_Size_X"]),1)*max(int(r.get("Grid_Size_Y",1)),1)//max(int(r.get("Workgroup_Size_Y",1)),1)
    name=r["Kernel_Name"].replace("void mp::","").split("(")[0][:44]
    print("%8.1f us  +%6.1f gap  wgs %6d  %s" % ((e-s)/1e3,(s-prev_end)/1e3,g,name))
    prev_end=max(prev_end,e)
    by[name][0]+=e-s; by[name][1]+=1
print()
for k,(t,c) in sorted(by.items(), key=lambda kv:-kv[1][0]): print("%-46s %3d launches %8.1f us" % (k,c,t/1e3))
PY
rm -rf $out/trace
