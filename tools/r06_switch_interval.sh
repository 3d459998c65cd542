cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06k
for v in 0 0.001 0.0002 0.00005; do
MONOPORT_STAGE_SWITCH_INTERVAL=$v timeout 600 python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench_dropin
res = bench_dropin.soak("cuda:0", 6.0, [17,33,65,129,257], window_s=2.0)
print("switch interval $v: soak %.1f recon/s (windows %s), steady p50 %.1f p99 %.1f; by in flight %s" % (res["value"], [round(w["value"],1) for w in res["windows"]], res["latency_ms_after_first_window"]["p50"], res["latency_ms_after_first_window"]["p99"], {k:(round(x["value"],1), round(x["latency_ms"]["p50"],2)) for k,x in res["latency_by_frames_in_flight"].items()}))
PY
done
