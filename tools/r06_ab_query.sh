cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for l in product libmp_oldquery.so; do
timeout 300 python tools/ab_lib.py $l --steps 20 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$l] value %.1f frac %.4f passes %s' % (d['value'], d['roofline']['frac'], [round(x,3) for x in d['passes']['ms_per_step_all']]))"
done; done
