#!/bin/bash
# Round 4 gpurun command lines, one script: tools/r04_run.sh STEP [OUTDIR]
#   ws      quick hang check + A/B probe of the wave-specialised table kernel, full GPU suite, bench A/B
#   tests   full GPU suite only
#   bench   default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
step=${1:-ws}
out=$R/gpurun_out/${2:-r04_$step}; mkdir -p $out
cd $R
bench_line() {  # $1 = json file, $2 = label
python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; ex=r.get("executed") or {}
print("[%s]" % sys.argv[2], "value", round(d["value"],2), "ms/step", round(d["ms_per_step"],3), "passes", [round(x,3) for x in d.get("passes",{}).get("ms_per_step_all",[])],
      "frac", round(r["frac"],4), "executed", round(ex.get("frac",0),4), "recon/frame", round(d["breakdown"]["recon_vertices_render_ms_per_frame_batched"],3),
      "enc", round(d["breakdown"]["encoder_ms_per_frame"],3))
PY
}
run_tests() {
  timeout 1500 python -m pytest tests -q -m gpu > $out/tests.log 2>&1
  echo "pytest rc=$?" >> $out/tests.log
  grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/tests.log | tail -15
}
case $step in
ws)
  timeout 180 python tools/tab_ws_probe.py quick > $out/probe_quick.txt 2>&1; rc=$?
  tail -5 $out/probe_quick.txt
  if [ $rc -ne 0 ]; then echo "quick probe rc=$rc -- stopping"; exit 1; fi
  timeout 300 python tools/tab_ws_probe.py > $out/probe.txt 2>&1; echo "probe rc=$?"; tail -6 $out/probe.txt
  run_tests
  for v in ws v1; do
    MONOPORT_TAB_KERNEL=$v timeout 600 python bench.py --no-extras --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err
    bench_line $out/bench_$v.json $v
  done
  ;;
tests) run_tests ;;
bench)
  timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.err
  bench_line $out/bench.json default ;;
esac
