#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r03d}; mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_encoder_dataflow_gpu.py tests/test_conv_gpu.py -q -m gpu > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR" $out/tests.log | tail -12
timeout 300 python tools/enc_latency.py f32 1 10 > $out/enc_latency.log 2>&1
tail -4 $out/enc_latency.log
timeout 300 python tools/conv_epi_probe.py 1 10 2>&1 | grep -v amdgpu.ids > $out/epi_probe.log
cat $out/epi_probe.log
MODES=auto timeout 600 python tools/conv_bench.py 1 10 2>&1 | grep -v amdgpu.ids > $out/conv_bench.log
cat $out/conv_bench.log
