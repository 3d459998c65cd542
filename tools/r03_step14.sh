#!/bin/bash
# layer-0 tables in the pipeline: full GPU suite, bench with and without
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r04k}; mkdir -p $out
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|Error" $out/tests.log | tail -8
for flag in "" "--no-skip-table" ""; do
timeout 600 python bench.py --no-extras --no-cpu-baseline $flag > $out/bench_ab.json 2> $out/bench.err
python - $out/bench_ab.json "$flag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ex=d["roofline"].get("executed") or {}
print("flag [%s]" % sys.argv[2], "value", round(d["value"],2), "ms/step", round(d["ms_per_step"],3), "passes", [round(x,3) for x in d.get("passes",{}).get("ms_per_step_all",[])], "frac(algorithmic)", round(d["roofline"]["frac"],4), "frac(executed)", round(ex.get("frac",0),4), "recon/frame", round(d["breakdown"]["recon_vertices_render_ms_per_frame_batched"],3))
PY
done
tail -3 $out/bench.err
