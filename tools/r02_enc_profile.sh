#!/bin/bash
# kernel trace + stats of the netG encoder alone (batch $2, default 5; eager, one stream) -- gpurun
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r02x}; mkdir -p $out
B=${2:-5}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/tools/enc_profile.py $B > $out/enc_prof.log 2>&1
cd $R
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python - "$f" "$B" <<'PY' | tee $out/encoder_kernel_stats_b$B.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
b=int(sys.argv[2]); frames=12*b
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("netG encoder alone, batch %d, 12 eager passes: total kernel time %.3f ms = %.3f ms/frame" % (b, tot/1e6, tot/1e6/frames))
for r in rows[:24]:
    print("%-60s calls %5s  total %8.3f ms  avg %8.1f us  %5.1f %%  = %.3f ms/frame" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot, float(r["TotalDurationNs"])/1e6/frames))
PY
rm -rf $out/trace
