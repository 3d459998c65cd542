#!/bin/bash
# the two PMC passes behind roofline.traffic at bench.py's default 16 frames per launch (counters only,
# separate runs, as MI355X_MICROARCH.md prescribes)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r04r}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python $R/tools/traffic_probe.py run > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python $R/tools/traffic_probe.py run > $out/pmc_write.log 2>&1
cd $R
python tools/traffic_probe.py parse $out/pmc_fetch $out/pmc_write $out/r03_query_traffic.json > $out/traffic_parse.log 2>&1
tail -3 $out/pmc_fetch.log; tail -12 $out/traffic_parse.log
rm -rf $out/pmc_fetch $out/pmc_write
