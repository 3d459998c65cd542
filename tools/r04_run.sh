#!/bin/bash
# Round 4 gpurun command lines (HISTORICAL: the ws / wsab / wspmc / order / abtraffic steps need the side builds and
# MONOPORT_TAB_KERNEL=v1 of commit 78cb450; round 5 is tools/r05_run.sh): tools/r04_run.sh STEP [OUTDIR]
#   ws      quick hang check + A/B probe of the wave-specialised table kernel, full GPU suite, bench A/B
#   tests   full GPU suite only
#   bench   default bench line        bench20  the driver's invocation        benchq  headline only
#   levels  headline bench + per-level roof fractions of its roofline leg (tools/launch_levels.py)
#   profhl  rocprofv3 kernel shares of the headline configuration alone       prof  kernel stats of the full bench for profiles/
#   traffic the --pmc FETCH_SIZE / WRITE_SIZE passes behind roofline.traffic  wspmc  counters of the table kernel
#   wsab    side builds of tools/ablate.py ($ABLATE) next to the product      conv / color / dropin / tab16 / octree / order / shapes*
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
  timeout 300 python tools/tab_ws_probe.py > $out/probe.txt 2>&1; echo "probe rc=$?"; tail -5 $out/probe.txt
  for name in $ABLATE; do
    MONOPORT_ABLATE=$name timeout 200 python tools/tab_ws_probe.py 2>&1 | tail -2 | tee -a $out/ablate.txt
  done
  ;;
wspmc)  # counters of the table query kernel: product (ws), round 3's (v1), and side builds in $ABLATE
  cd /tmp && export TMPDIR=/tmp
  for v in ws v1 $ABLATE; do
    if [ $v = ws ] || [ $v = v1 ]; then export MONOPORT_TAB_KERNEL=$v; unset MONOPORT_ABLATE; else export MONOPORT_ABLATE=$v; unset MONOPORT_TAB_KERNEL; fi
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_a_$v -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_a_$v.log 2>&1
    rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc_b_$v -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_b_$v.log 2>&1
    rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $out/pmc_c_$v -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_c_$v.log 2>&1
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_d_$v -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_d_$v.log 2>&1
    for ps in a b c d; do echo "== $v pass $ps"; python $R/tools/pmc_summary.py $out/pmc_${ps}_$v | grep -v skip_table | tail -1; done >> $out/pmc_summary.txt 2>&1
    rm -rf $out/pmc_a_$v $out/pmc_b_$v $out/pmc_c_$v $out/pmc_d_$v
  done
  unset MONOPORT_ABLATE MONOPORT_TAB_KERNEL
  cat $out/pmc_summary.txt
  ;;
traffic)  # the PMC passes behind roofline.traffic at 10 and 16 frames per launch (counters only, separate runs)
  cd /tmp && export TMPDIR=/tmp
  for b in 10 16 20; do
    export MONOPORT_TRAFFIC_BATCH=$b
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$b -- python $R/tools/traffic_probe.py run > $out/pmc_fetch_$b.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$b -- python $R/tools/traffic_probe.py run > $out/pmc_write_$b.log 2>&1
    (cd $R && python tools/traffic_probe.py parse $out/pmc_fetch_$b $out/pmc_write_$b $out/traffic_$b.json > $out/traffic_parse_$b.log 2>&1)
    tail -2 $out/pmc_fetch_$b.log | cut -c1-200
    rm -rf $out/pmc_fetch_$b $out/pmc_write_$b
  done
  unset MONOPORT_TRAFFIC_BATCH
  cd $R
  python - $out <<'PY'
import json,sys,os
out=sys.argv[1]
by={}
for b in (10,16,20):
    d=json.load(open(os.path.join(out,"traffic_%d.json"%b)))
    by[str(b)]=d
merged={"source":"tools/r04_run.sh traffic: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/traffic_probe.py run at slot batches of 10, 16 and 20 frames (bench.py --steps 20 --batch 10 / default --steps 48 / --steps 20: launches of 16 + 4 frames)","by_slot_batch":by}
json.dump(merged,open(os.path.join(out,"r04_query_traffic.json"),"w"),indent=1)
for b,d in by.items():
    print(b,"frames/launch: avg %.3f GB per launch; per level (GB):"%(d["bytes_per_launch_avg"]/1e9),[round(x/1e9,3) for x in d["bytes_per_level_launch"]], "WRITE KB", [round(x) for x in d["WRITE_SIZE_per_level"]])
PY
  ;;
prof)  # kernel trace + stats of the default bench with the per-launch point counts of the roofline leg
  cd /tmp && export TMPDIR=/tmp
  CMD="python bench.py --warmup 5 --no-alt --no-dropin --no-cpu-baseline"
  MONOPORT_BENCH_LAUNCH_LOG=$out/launch_log.json rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --warmup 5 --no-alt --no-dropin --no-cpu-baseline > $out/bench_prof.log 2>&1
  cd $R
  python tools/profile_summary.py $out/trace $out/r04_bench "$CMD" 10 $out/launch_log.json > $out/summary.log 2>&1
  tail -1 $out/bench_prof.log | cut -c1-300; cat $out/summary.log | tail -30
  rm -rf $out/trace
  ;;
profhl)  # kernel stats of the headline configuration alone (no extras): where a frame's GPU time goes
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --no-extras --no-cpu-baseline > $out/bench.log 2>&1
  cd $R
  tail -1 $out/bench.log | cut -c1-200
  f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%6.2f %%  %7d calls  avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
  cp "$f" $out/headline_kernel_stats.csv; rm -rf $out/trace
  ;;
order)  # point-list order of the octree (y-major default | z-major) x tile order of the table kernel: time + traffic
  for ord in y z; do
    export MONOPORT_OCTREE_ORDER=$ord
    echo "== octree order $ord"
    timeout 300 python tools/tab_ws_probe.py 2>&1 | tail -1
    MONOPORT_ABLATE=wsflat timeout 300 python tools/tab_ws_probe.py 2>&1 | tail -1
    (cd /tmp && export TMPDIR=/tmp MONOPORT_TRAFFIC_BATCH=16
     rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf_$ord -- python $R/tools/traffic_probe.py run > $out/pf_$ord.log 2>&1
     rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw_$ord -- python $R/tools/traffic_probe.py run > $out/pw_$ord.log 2>&1)
    python tools/traffic_probe.py parse $out/pf_$ord $out/pw_$ord $out/traffic_$ord.json | grep -A6 bytes_per_level_launch | tr -d '\n '; echo
    rm -rf $out/pf_$ord $out/pw_$ord
  done
  unset MONOPORT_OCTREE_ORDER
  ;;
abtraffic)  # time + level traffic (16 frames per launch) of side builds in $ABLATE next to the product
  for name in product $ABLATE; do
    if [ $name = product ]; then unset MONOPORT_ABLATE; else export MONOPORT_ABLATE=$name; fi
    echo "== $name"
    timeout 300 python tools/tab_ws_probe.py 2>&1 | tail -1
    (cd /tmp && export TMPDIR=/tmp MONOPORT_TRAFFIC_BATCH=16
     rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf_$name -- python $R/tools/traffic_probe.py run > $out/pf_$name.log 2>&1
     rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw_$name -- python $R/tools/traffic_probe.py run > $out/pw_$name.log 2>&1)
    python tools/traffic_probe.py parse $out/pf_$name $out/pw_$name $out/traffic_$name.json | grep -A6 bytes_per_level_launch | tr -d '\n '; echo
    rm -rf $out/pf_$name $out/pw_$name
  done
  unset MONOPORT_ABLATE
  ;;
octree)  # one frame's octree kernels under rocprofv3, by point-list order
  cd /tmp && export TMPDIR=/tmp
  for ord in y z; do
    MONOPORT_OCTREE_ORDER=$ord rocprofv3 --kernel-trace --stats --output-format csv -d $out/oct_$ord -- python $R/tools/octree_probe.py > $out/oct_$ord.log 2>&1
    echo "== order $ord: $(grep recon $out/oct_$ord.log)"
    python - $out/oct_$ord <<'PY'
import csv,glob,sys,os
for p in glob.glob(os.path.join(sys.argv[1],"**","*kernel_stats.csv"),recursive=True):
    for r in csv.DictReader(open(p)):
        if any(k in r["Name"] for k in ("select_compact","upsample","iota","query")):
            print("   %-46s calls %4s avg %8.1f us" % (r["Name"][:46], r["Calls"], float(r["AverageNs"])/1e3))
PY
    rm -rf $out/oct_$ord
  done
  ;;
dropin)
  timeout 900 python bench.py --mode dropin > $out/dropin.json 2> $out/dropin.err; tail -c 300 $out/dropin.err
  python - $out/dropin.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("coalesced", round(d["value"],1), d["passes"], "| per-frame stages", round(d["per_frame_stages"]["value"],1), d["per_frame_stages"]["passes"], "| latency", round(d["latency_ms_single_frame"],2))
PY
  ;;
tab16)  # the f16 query kernels through the skip table: hang check, probe, their tests
  timeout 240 python tools/tab16_probe.py quick > $out/probe_quick.txt 2>&1; rc=$?
  tail -8 $out/probe_quick.txt
  if [ $rc -ne 0 ]; then echo "quick probe rc=$rc -- stopping"; exit 1; fi
  timeout 400 python tools/tab16_probe.py > $out/probe.txt 2>&1; echo "probe rc=$?"; tail -8 $out/probe.txt
  timeout 600 python -m pytest tests/test_query_gpu.py -q -m gpu -k "f16 or fp16" 2>&1 | tail -8
  ;;
shapes)  # headline by slot layout / encoder launch mode
  for flags in "" "--no-graph" "--depth 4 --batch 8 --steps 64" "--depth 6 --batch 8 --steps 96" "--depth 2 --batch 16 --steps 64" "--depth 4 --batch 8 --steps 64 --no-graph" "--depth 6 --batch 4 --steps 96"; do
    timeout 600 python bench.py --no-extras --no-cpu-baseline $flags > $out/b.json 2> $out/b.err
    bench_line $out/b.json "$flags"
  done
  ;;
conv)  # the encoder's convolutions per shape (batch 1 and 16) + the encoder tests
  MODES=auto timeout 600 python tools/conv_bench.py 1 16 > $out/conv_bench.txt 2>&1; tail -45 $out/conv_bench.txt | cut -c1-110
  timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_encoder_dataflow_gpu.py -q -m gpu 2>&1 | tail -4
  ;;
shapes20)  # the driver's --steps 20 by slot layout
  for flags in "" "--batch 5" "--batch 4" "--batch 20" "--batch 10 --depth 2" "--batch 5 --depth 4" "--batch 2"; do
    timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline $flags > $out/b.json 2> $out/b.err
    bench_line $out/b.json "$flags"
  done
  ;;
color)  # configs[2] + the convk tests
  timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_encoder_dataflow_gpu.py tests/test_dropin_gpu.py -q -m gpu -k "convk or netc or encoder or stem" 2>&1 | tail -3
  timeout 600 python bench.py --with-color --no-extras --no-cpu-baseline > $out/color.json 2> $out/color.err; tail -c 200 $out/color.err
  bench_line $out/color.json with-color
  timeout 300 python tools/netc_encoder_probe.py 2>&1 | tail -6
  ;;
levels)  # the headline bench + the per-level fractions of its roofline leg
  MONOPORT_BENCH_LAUNCH_LOG=$out/launch_log.json timeout 600 python bench.py --no-extras --no-cpu-baseline > $out/bench.json 2> $out/bench.err
  bench_line $out/bench.json headline
  python tools/launch_levels.py $out/launch_log.json | tee $out/levels.txt
  ;;
tests) run_tests ;;
bench)
  timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.err
  bench_line $out/bench.json default ;;
bench20)  # the driver's invocation
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench20.json 2> $out/bench20.err; tail -c 300 $out/bench20.err
  bench_line $out/bench20.json steps20
  python - $out/bench20.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("traffic", r["traffic"], "step", r["step"])
for k in ("plain_query_path","in_flight_8","alt_precision","with_color","levels6_f16w","dropin","mesh","cpu_baseline"):
    v=d.get(k)
    if v: print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","roofline_frac","passes","latency_ms_single_frame","mesh_ms","per_frame_stages")})
PY
  ;;
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
