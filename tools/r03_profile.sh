#!/bin/bash
# Round-3 profile collection on the GPU box (through gpurun): kernel trace + stats of the default
# bench with the per-launch point counts of the roofline leg; kernel stats of the mesh leg.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r04q}
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python bench.py --warmup 5 --no-alt --no-dropin --no-cpu-baseline"
MONOPORT_BENCH_LAUNCH_LOG=$out/launch_log.json rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --warmup 5 --no-alt --no-dropin --no-cpu-baseline > $out/bench_prof.log 2>&1
cd $R
python tools/profile_summary.py $out/trace $out/r03_bench "$CMD" 10 $out/launch_log.json > $out/summary.log 2>&1
tail -1 $out/bench_prof.log | cut -c1-400; cat $out/summary.log
rm -rf $out/trace/*/*.db 2>/dev/null
find $out/trace -name "*kernel_trace.csv" -size +20M -delete
