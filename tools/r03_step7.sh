#!/bin/bash
# where to put the gate between the 32- and 64-point query kernels: the bench value at several gates
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r03w}; mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_query_gpu.py tests/test_recon_gpu.py -q -m gpu -x > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/tests.log | tail -5
timeout 600 python tools/small_tile_probe.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee $out/small_tile_probe.log
for gate in 512 1 512 1 2048 8192 0; do
  MONOPORT_QUERY_SMALL_TILES=$gate timeout 600 python bench.py --no-extras --no-cpu-baseline > $out/bench_gate$gate.json 2> $out/bench.err
  python - $out/bench_gate$gate.json $gate <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("gate", sys.argv[2], "value", round(d["value"],2), "ms/step", round(d["ms_per_step"],3), "passes", [round(x,3) for x in d.get("passes",{}).get("ms_per_step_all",[])], "frac", round(d["roofline"]["frac"],4))
PY
done
