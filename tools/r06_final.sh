cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r06y}; mkdir -p $out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 300 $out/bench_default.err
python - $out/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("default: value %.1f ms %.3f frac %.4f lfl %.4f enc %.3f/%.3f color %s l6 %s dropin %.1f per-frame %.1f trusted %.1f lat %.2f soak %.1f steps %d" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("like_for_like_frac", 0), d["breakdown"]["encoder_ms_per_frame"], d["breakdown"]["encoder_ms_batch1"],
    d.get("with_color", {}).get("value"), d.get("levels6_f16w", {}).get("value"), d["dropin"]["value"], d["dropin"]["per_frame_stages"]["value"],
    d["dropin"]["per_frame_stages_trusted"]["value"], d["dropin"]["latency_ms_single_frame"], d["dropin"]["soak"]["value"], d["steps"]))
PY
