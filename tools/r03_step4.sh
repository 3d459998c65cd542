#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r03j}; mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_encoder_dataflow_gpu.py tests/test_conv_gpu.py -q -m gpu > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/tests.log | tail -6
MODES=auto timeout 600 python tools/conv_bench.py 1 10 2>&1 | grep -v amdgpu.ids | grep "1x1\|per frame" | tee $out/conv_bench_1x1.log
for B in 1 10; do
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace$B -- python $R/tools/enc_profile.py $B > $out/enc_prof$B.log 2>&1
  cd $R
  f=$(find $out/trace$B -name "*kernel_stats.csv" | head -1)
  python - "$f" "$B" <<'PY' > $out/encoder_kernel_stats_b$B.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
b=int(sys.argv[2]); frames=12*b
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("netG encoder alone (hand-over dataflow), batch %d, 12 eager passes: total kernel time %.3f ms = %.3f ms/frame" % (b, tot/1e6, tot/1e6/frames))
for r in rows[:30]:
    print("%-64s calls %5s  total %8.3f ms  avg %8.1f us  %5.1f %%  = %.3f ms/frame" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot, float(r["TotalDurationNs"])/1e6/frames))
PY
  rm -rf $out/trace$B
  head -22 $out/encoder_kernel_stats_b$B.txt
done
