#!/bin/bash
# PMC counters of the skip-table kernels (counters only, separate passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r04s}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_a -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $out/pmc_b -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_b.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/pmc_c -- python $R/tools/skip_table_pmc_probe.py > $out/pmc_c.log 2>&1
cd $R
{ python tools/pmc_summary.py $out/pmc_a | tail -2; python tools/pmc_summary.py $out/pmc_b | tail -2; python tools/pmc_summary.py $out/pmc_c | tail -2; } > $out/pmc_tab_summary.txt 2>&1
cat $out/pmc_tab_summary.txt; tail -2 $out/pmc_c.log
rm -rf $out/pmc_a $out/pmc_b $out/pmc_c
