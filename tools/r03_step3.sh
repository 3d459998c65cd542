#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r03i}; mkdir -p $out
cd $R
timeout 1200 python -m pytest tests/test_baseline_size_gpu.py tests/test_recon_gpu.py tests/test_encoder_dataflow_gpu.py tests/test_dropin_gpu.py tests/test_conv_gpu.py -q -m gpu -s > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|pipeline257" $out/tests.log | tail -14
timeout 300 python tools/dropin_latency_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/dropin_latency.txt
timeout 300 python tools/enc_latency.py f32 1 10 2>&1 | grep -v amdgpu.ids | tee $out/enc_latency.log
