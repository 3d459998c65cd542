#!/bin/bash
# Round-2 profile collection on the GPU box (through gpurun): kernel trace + stats of the default
# bench, then the two PMC passes behind roofline.traffic (separate runs, counters only).
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r02c}
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --steps 20 --warmup 5 --no-alt --no-dropin --no-cpu-baseline > $out/bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python $R/tools/traffic_probe.py run > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python $R/tools/traffic_probe.py run > $out/pmc_write.log 2>&1
cd $R
python tools/traffic_probe.py parse $out/pmc_fetch $out/pmc_write $out/r02_query_traffic.json > $out/traffic_parse.log 2>&1
python tools/profile_summary.py $out/trace $out/r02_bench "python bench.py --steps 20 --warmup 5 --no-alt --no-dropin --no-cpu-baseline" 20 > $out/summary.log 2>&1
tail -3 $out/bench_prof.log | cut -c1-600; tail -5 $out/traffic_parse.log; cat $out/summary.log
# keep the merge small: the raw traces are large
rm -rf $out/trace/*/*.db $out/pmc_fetch $out/pmc_write 2>/dev/null
find $out/trace -name "*kernel_trace.csv" -size +20M -delete
