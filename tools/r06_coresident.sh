cd $GRAFT_REPO_ROOT
out=gpurun_out/r06h; mkdir -p $out
run() { # label, env, flags
  label=$1; shift; envs=$1; shift
  env $envs timeout 600 python bench.py --no-extras --no-cpu-baseline "$@" > $out/b_$label.json 2> $out/b_$label.err
  python - $out/b_$label.json "$label $envs $*" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("[%s]"%sys.argv[2], "value %.1f ms/step %.3f passes %s frac %.4f" % (d["value"], d["ms_per_step"], [round(x,3) for x in d["passes"]["ms_per_step_all"]], d["roofline"]["frac"]))
except Exception as e:
    print("[%s] failed %r" % (sys.argv[2], e))
PY
}
run base20 "A=1" --steps 20 --warmup 5
run base2x10 "A=1" --steps 20 --warmup 5 --batch 10 --depth 2
run wg1_2x10 "MONOPORT_QUERY_WGS_PER_CU=1" --steps 20 --warmup 5 --batch 10 --depth 2
run wg1_4x5 "MONOPORT_QUERY_WGS_PER_CU=1" --steps 20 --warmup 5 --batch 5 --depth 4
run base_4x5 "A=1" --steps 20 --warmup 5 --batch 5 --depth 4
run wg1_1x20 "MONOPORT_QUERY_WGS_PER_CU=1" --steps 20 --warmup 5
run wg1_40_4x10 "MONOPORT_QUERY_WGS_PER_CU=1" --steps 40 --warmup 5 --batch 10 --depth 4
run base_40_4x10 "A=1" --steps 40 --warmup 5 --batch 10 --depth 4
