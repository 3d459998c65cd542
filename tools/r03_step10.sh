#!/bin/bash
# NR = 1 convolution kernels on two accumulators: parity, per-shape times, encoder latency
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r04a}; mkdir -p $out
cd $R
timeout 1200 python -m pytest tests/test_encoder_dataflow_gpu.py tests/test_conv_gpu.py tests/test_dropin_gpu.py -q -m gpu -x > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/tests.log | tail -6
MODES=auto,large1,large2 timeout 600 python tools/conv_bench.py 1 16 2>&1 | grep -v amdgpu.ids | tee $out/conv_bench.log
timeout 300 python tools/enc_latency.py f32 1 16 2>&1 | grep -v amdgpu.ids | tee $out/enc_latency.log
