#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r03k}; mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_encoder_dataflow_gpu.py tests/test_conv_gpu.py tests/test_dropin_gpu.py -q -m gpu -s > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|full-coverage|netC.filter dataflow" $out/tests.log | tail -10
timeout 300 python tools/enc_latency.py f32 1 10 2>&1 | grep -v amdgpu.ids | tee $out/enc_latency.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
python - $out/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "enc/frame", d["breakdown"]["encoder_ms_per_frame"], "enc b1", d["breakdown"]["encoder_ms_batch1"], "recon", d["breakdown"]["recon_vertices_render_ms_per_frame_batched"])
print("roofline frac", d["roofline"]["frac"], "alt", d["alt_precision"]["value"], "color", d["with_color"]["value"], "513", d["levels6_f16w"]["value"], "dropin", d["dropin"]["value"], d["dropin"]["passes"], d["dropin"]["latency_ms_single_frame"], "inflight8", d["in_flight_8"]["value"])
PY
