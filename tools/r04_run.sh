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
wsab)  # side builds of tools/ablate.py (names in $ABLATE) next to the product
  timeout 180 python tools/tab_ws_probe.py quick > $out/probe_quick.txt 2>&1; rc=$?
  tail -3 $out/probe_quick.txt
  if [ $rc -ne 0 ]; then echo "quick probe rc=$rc -- stopping"; exit 1; fi
  timeout 300 python tools/tab_ws_probe.py > $out/probe.txt 2>&1; echo "probe rc=$?"; tail -3 $out/probe.txt
  for name in $ABLATE; do
    MONOPORT_ABLATE=$name timeout 200 python tools/tab_ws_probe.py 2>&1 | tail -1 | tee -a $out/ablate.txt
  done
  ;;
wspmc)  # counters of the table query kernel: product (ws), round 3's (v1), and side builds in $ABLATE
  cd /tmp && export TMPDIR=/tmp
  for v in ws v1 $ABLATE; do
    if [ $v = ws ] || [ $v = v1 ]; then export MONOPORT_TAB_KERNEL=$v; unset MONOPORT_ABLATE; else export MONOPORT_ABLATE=$v; unset MONOPORT_TAB_KERNEL; fi
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_a_$v -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_a_$v.log 2>&1
    rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc_b_$v -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_b_$v.log 2>&1
    rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $out/pmc_c_$v -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_c_$v.log 2>&1
    for ps in a b c; do echo "== $v pass $ps"; python $R/tools/pmc_summary.py $out/pmc_${ps}_$v | grep -v skip_table | tail -1; tail -1 $out/pmc_${ps}_$v.log | cut -c1-300; done >> $out/pmc_summary.txt 2>&1
    rm -rf $out/pmc_a_$v $out/pmc_b_$v $out/pmc_c_$v
  done
  unset MONOPORT_ABLATE MONOPORT_TAB_KERNEL
  cat $out/pmc_summary.txt
  ;;
tests) run_tests ;;
bench)
  timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.err
  bench_line $out/bench.json default ;;
benchq)  # headline only, no extras
  timeout 600 python bench.py --no-extras --no-cpu-baseline > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.err
  bench_line $out/bench.json quick
  python - $out/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps(d["roofline"])[:1500])
PY
  ;;
esac
