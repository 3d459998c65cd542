#!/bin/bash
# kernel trace + stats of the opt-in f16x3 configuration (gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r02k}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --steps 20 --warmup 5 --precision f16x3 --no-dropin --no-cpu-baseline > $out/bench_prof.log 2>&1
cd $R
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
head -45 $f > $out/r02_bench_f16x3_kernel_stats.csv
cut -c1-110 $out/r02_bench_f16x3_kernel_stats.csv | head -40
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time ms", tot/1e6)
PY
rm -rf $out/trace
