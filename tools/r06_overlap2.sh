cd $GRAFT_REPO_ROOT; out=gpurun_out/r06r; mkdir -p $out
run() { label=$1; shift; env "$@" timeout 300 python tools/per_frame_overlap_probe.py $label 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $out/overlap.txt; }
for rep in 1 2; do
run base A=1
run mult2 MONOPORT_QUERY_GRID_MULT=2
run mult4 MONOPORT_QUERY_GRID_MULT=4
run mult8 MONOPORT_QUERY_GRID_MULT=8
run mult16 MONOPORT_QUERY_GRID_MULT=16
done
for m in 1 4; do
MONOPORT_QUERY_GRID_MULT=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[headline mult $m] value %.1f frac %.4f passes %s' % (d['value'], d['roofline']['frac'], d['passes']['ms_per_step_all']))" | tee -a $out/overlap.txt
done
