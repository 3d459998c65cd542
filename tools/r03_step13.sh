#!/bin/bash
# skip-table path: parity and timing
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r04i}; mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_query_gpu.py -q -m gpu -x -s -k "skip_table or small_tile" > $out/tests.log 2>&1
echo "pytest rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|table-f64|Error|assert " $out/tests.log | tail -12
timeout 600 python tools/skip_table_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/skip_table_probe.log
