#!/bin/bash
# PMC passes of the f16x3 query kernel on a 1 M-point launch (gpurun): MFMA busy, wait buckets, LDS
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r02p}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MP_PROBE_PREC=${2:-f16x3}
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_a -- python $R/tools/pmc_probe.py 1048576 > $out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $out/pmc_b -- python $R/tools/pmc_probe.py 1048576 > $out/pmc_b.log 2>&1
cd $R
python tools/pmc_summary.py $out/pmc_a | tail -1 > $out/pmc_${MP_PROBE_PREC}_summary.txt
python tools/pmc_summary.py $out/pmc_b | tail -1 >> $out/pmc_${MP_PROBE_PREC}_summary.txt
cat $out/pmc_${MP_PROBE_PREC}_summary.txt; tail -3 $out/pmc_b.log
rm -rf $out/pmc_a $out/pmc_b
