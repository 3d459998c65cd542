#!/bin/bash
# gate at 2048 tiles: full GPU suite, MFMA peak probe (constant / random operands, shader clock),
# tile-size sweep, drop-in latency, default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r03x}; mkdir -p $out
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/tests.log | tail -8
timeout 120 tools/probes/bin/mfma_peak_probe 2>&1 | tee $out/mfma_peak_probe.log
timeout 600 python tools/small_tile_probe.py 2>&1 | grep -v amdgpu.ids > $out/small_tile_probe.log; tail -4 $out/small_tile_probe.log
timeout 600 python tools/dropin_latency_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/dropin_latency.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
python - $out/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "enc/frame", d["breakdown"]["encoder_ms_per_frame"], "enc b1", d["breakdown"]["encoder_ms_batch1"], "recon", d["breakdown"]["recon_vertices_render_ms_per_frame_batched"])
print("roofline frac", d["roofline"]["frac"], "alt", d["alt_precision"]["value"], "color", d["with_color"]["value"], "513", d["levels6_f16w"]["value"], "dropin", d["dropin"]["value"], d["dropin"]["passes"], d["dropin"]["latency_ms_single_frame"], "inflight8", d["in_flight_8"]["value"])
PY
