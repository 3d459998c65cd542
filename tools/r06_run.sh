#!/bin/bash
# Round 6 gpurun command lines, one script: tools/r06_run.sh STEP [OUTDIR]
#   first    smoke + full GPU suite + the driver's bench invocation + per-level roof fractions (20 and 24 frames per launch)
#   tests    full GPU suite only            tests_k  pytest -k "$K" only
#   bench    default bench line             bench20  the driver's invocation        benchq  headline only ($FLAGS)
#   levels / levels20   headline bench + per-level roof fractions of its roofline leg (tools/launch_levels.py)
#   prof20   rocprofv3 --kernel-trace --stats of the driver's invocation, summarised for profiles/
#   traffic  the --pmc FETCH_SIZE / WRITE_SIZE passes behind roofline.traffic at 16 / 20 / 24 frames per launch
#   pmc      counters of the table query kernel (MFMA busy, INSTS_MFMA, L2 hits)
#   dropin   bench.py --mode dropin         shapes  headline by slot layout         f16w  configs[4] alone
R=${GRAFT_REPO_ROOT:-/root/repo}
step=${1:-first}
out=$R/gpurun_out/${2:-r06_$step}; mkdir -p $out
cd $R
bench_line() {  # $1 = json file, $2 = label
python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("[%s] no JSON line: %r" % (sys.argv[2], e)); sys.exit(0)
r=d["roofline"]; st=(r.get("step") or {}).get("executed") or {}
print("[%s]" % sys.argv[2], "value", round(d["value"],2), "ms/step", round(d["ms_per_step"],3), "passes", [round(x,3) for x in d.get("passes",{}).get("ms_per_step_all",[])],
      "frac", round(r["frac"],4), "like-for-like", r.get("like_for_like_frac"), "step", round(st.get("frac",0),4), "frames/launch", r.get("frames_per_launch"),
      "recon/frame", round(d["breakdown"]["recon_vertices_render_ms_per_frame_batched"],3), "enc", round(d["breakdown"]["encoder_ms_per_frame"],3),
      "enc(as run)", d["breakdown"].get("encoder_ms_per_frame_as_run"), "enc@1", round(d["breakdown"]["encoder_ms_batch1"],3))
if r.get("sustained"): print("    sustained", {k:(round(v,4) if isinstance(v,float) else v) for k,v in r["sustained"].items() if k != "note"})
for k in ("plain_query_path","two_slot_submissions","in_flight_8","alt_precision","with_color","levels6_f16w","mesh","cpu_baseline","cpu_baseline_reference_ops"):
    v=d.get(k)
    if v: print("   ", k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","roofline_frac","roofline_frac_netG_query","roofline_frac_netC_query","ms_per_step","mesh_ms","iou_vs_f32_volume")})
f=d.get("final_level_rules")
if f:
    print("    final_level_rules: headline points/level", [round(x) for x in f["headline_points_per_level"]])
    for k in ("upstream","interpolate"):
        v=f[k]; print("      %-11s value %.1f points/level %s differing voxels %d of %d inside, IoU %.5f" % (k, v["value"], [round(x) for x in v["points_per_level"]], v["thresholded_voxels_differing_from_dilate3"], v["inside_voxels"], v["iou_vs_dilate3"]))
dr=d.get("dropin")
if dr:
    print("    dropin: coalesced %.1f | per-frame stages (validate=always) %.1f | per-frame trusted %.1f | latency %.2f ms" % (dr["value"], dr["per_frame_stages"]["value"], dr["per_frame_stages_trusted"]["value"], dr["latency_ms_single_frame"]))
    sk=dr.get("soak")
    if sk: print("    soak: %.1f recon/s over %.1f s, %d frames (%d None), latency p50 %.1f p99 %.1f max %.1f ms, flat %s; by frames in flight: %s" % (sk["value"], sk["seconds"], sk["frames"], sk["none_frames"], sk["latency_ms_after_first_window"]["p50"], sk["latency_ms_after_first_window"]["p99"], sk["latency_ms_after_first_window"]["max"], sk["flat_after_warmup"], {k:(round(v["value"],1), round(v["latency_ms"]["p50"],1), round(v["latency_ms"]["p99"],1)) for k,v in sk.get("latency_by_frames_in_flight",{}).items()}))
PY
}
run_tests() {
  timeout 1800 python -X faulthandler -m pytest tests -q -m gpu ${K:+-k "$K"} -s > $out/tests.log 2>&1
  echo "pytest rc=$?" >> $out/tests.log
  grep -E "passed|failed|^FAILED|^ERROR|rc=|configs\[2\]|colour chain|final_level=|pipeline257_color|513\^3|pipeline513" $out/tests.log | tail -40
}
levels() {  # $1 = label, rest = bench flags
  label=$1; shift
  MONOPORT_BENCH_LAUNCH_LOG=$out/launch_log_$label.json timeout 600 python bench.py --no-extras --no-cpu-baseline "$@" > $out/bench_$label.json 2> $out/bench_$label.err
  tail -c 300 $out/bench_$label.err
  bench_line $out/bench_$label.json $label
  python tools/launch_levels.py $out/launch_log_$label.json | tee $out/levels_$label.txt
}
case $step in
first)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  run_tests
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench20.json 2> $out/bench20.err; tail -c 300 $out/bench20.err
  bench_line $out/bench20.json steps20
  levels s20 --steps 20 --warmup 5
  levels s32
  ;;
tests|tests_k) run_tests ;;
bench)
  timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.err
  bench_line $out/bench.json default ;;
bench20)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench20.json 2> $out/bench20.err; tail -c 300 $out/bench20.err
  bench_line $out/bench20.json steps20 ;;
benchq)
  timeout 600 python bench.py --no-extras --no-cpu-baseline $FLAGS > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.err
  bench_line $out/bench.json "quick $FLAGS" ;;
levels) levels s32 ;;
levels20) levels s20 --steps 20 --warmup 5 ;;
pmc)  # counters of the shipped table query kernel on one 885 k-point lattice launch (counters only, separate passes)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_a -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_a.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $out/pmc_c -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_c.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_d -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_d.log 2>&1
  for ps in a c d; do echo "== pass $ps"; python $R/tools/pmc_summary.py $out/pmc_$ps | grep -v skip_table | tail -1; done > $out/pmc_summary.txt 2>&1
  rm -rf $out/pmc_a $out/pmc_c $out/pmc_d
  cat $out/pmc_summary.txt
  cd $R ;;
prof20)  # kernel trace + stats of the driver's invocation with the per-launch point counts of the roofline leg
  cd /tmp && export TMPDIR=/tmp
  CMD="python bench.py --gpus 1 --steps 20 --warmup 5"
  MONOPORT_BENCH_LAUNCH_LOG=$out/launch_log.json rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-dropin --no-cpu-baseline > $out/bench_prof.log 2>&1
  cd $R
  python tools/profile_summary.py $out/trace $out/r06_bench "$CMD --no-dropin --no-cpu-baseline" 20 $out/launch_log.json > $out/summary.log 2>&1
  tail -1 $out/bench_prof.log | cut -c1-300; tail -30 $out/summary.log
  rm -rf $out/trace
  ;;
traffic)
  cd /tmp && export TMPDIR=/tmp
  for b in 16 20 24 32; do
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
for b in (16,20,24,32):
    by[str(b)]=json.load(open(os.path.join(out,"traffic_%d.json"%b)))
merged={"source":"tools/r06_run.sh traffic: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/traffic_probe.py run at slot batches of 16, 20, 24 and 32 frames (bench.py --steps 48 / the driver's --steps 20 / --batch 24 / the default --steps 32): one fused-query launch per octree level","by_slot_batch":by}
json.dump(merged,open(os.path.join(out,"r06_query_traffic.json"),"w"),indent=1)
for b,d in by.items():
    print(b,"frames/launch: avg %.3f GB per launch; per level (GB):"%(d["bytes_per_launch_avg"]/1e9),[round(x/1e9,3) for x in d["bytes_per_level_launch"]], "WRITE KB", [round(x) for x in d["WRITE_SIZE_per_level"]])
PY
  ;;
traffic16)  # configs[4] (513^3, fp16 weights): memory-side bytes and L1 -> L2 requests of pifu_query16_kernel per level
  cd /tmp && export TMPDIR=/tmp MONOPORT_TRAFFIC_BATCH=16 MONOPORT_TRAFFIC_LEVELS=6 MONOPORT_TRAFFIC_PRECISION=f16w
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python $R/tools/traffic_probe.py run > $out/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python $R/tools/traffic_probe.py run > $out/pmc_write.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum --output-format csv -d $out/pmc_l2req -- python $R/tools/traffic_probe.py run > $out/pmc_l2req.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2hit -- python $R/tools/traffic_probe.py run > $out/pmc_l2hit.log 2>&1
  cd $R
  python tools/traffic_probe.py parse $out/pmc_fetch $out/pmc_write $out/r06_query_traffic_513_f16w.json > $out/parse.log 2>&1; tail -3 $out/parse.log
  python tools/traffic_probe.py counter $out/pmc_l2req TCP_TCC_READ_REQ_sum $out/l2req.json
  python tools/traffic_probe.py counter $out/pmc_l2hit TCC_HIT_sum $out/l2hit.json
  python tools/traffic_probe.py counter $out/pmc_l2hit TCC_MISS_sum $out/l2miss.json
  tail -2 $out/pmc_fetch.log | cut -c1-200
  rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_l2req $out/pmc_l2hit
  ;;
dropin)
  timeout 900 python bench.py --mode dropin > $out/dropin.json 2> $out/dropin.err; tail -c 300 $out/dropin.err
  python - $out/dropin.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("coalesced", round(d["value"],1), d["passes"], "| per-frame stages", round(d["per_frame_stages"]["value"],1), d["per_frame_stages"]["passes"], "| latency", round(d["latency_ms_single_frame"],2))
PY
  ;;
shapes)
  for flags in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --batch 10" "--steps 20 --warmup 5 --batch 10 --depth 2" "" "--batch 16" "--steps 96" "--steps 64 --batch 32 --depth 2"; do
    timeout 600 python bench.py --no-extras --no-cpu-baseline $flags > $out/b.json 2> $out/b.err
    bench_line $out/b.json "$flags"
  done
  ;;
f16w)
  timeout 600 python bench.py --levels 6 --precision f16w --no-extras --no-cpu-baseline $FLAGS > $out/f16w.json 2> $out/f16w.err; tail -c 300 $out/f16w.err
  bench_line $out/f16w.json "513 f16w $FLAGS" ;;
esac
