#!/bin/bash
# split accumulators in layers 0 / 3: parity, tile sweep, bench headline
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r03z}; mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_query_gpu.py tests/test_recon_gpu.py tests/test_baseline_size_gpu.py -q -m gpu -x > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|Error|assert" $out/tests.log | tail -8
timeout 600 python tools/small_tile_probe.py 2>&1 | grep -v amdgpu.ids > $out/small_tile_probe.log; cat $out/small_tile_probe.log
for gate in -1 1; do
MONOPORT_QUERY_SMALL_TILES=$gate timeout 600 python bench.py --no-extras --no-cpu-baseline > $out/bench_gate$gate.json 2> $out/bench.err
python - $out/bench_gate$gate.json $gate <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("gate", sys.argv[2], "value", round(d["value"],2), "ms/step", round(d["ms_per_step"],3), "passes", [round(x,3) for x in d.get("passes",{}).get("ms_per_step_all",[])], "frac", round(d["roofline"]["frac"],4))
PY
done
