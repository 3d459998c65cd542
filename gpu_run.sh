export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d /root/repo/gpurun_out/pmc1 -o q -- python /root/repo/tools/pmc_probe.py > /root/repo/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d /root/repo/gpurun_out/pmc2 -o q -- python /root/repo/tools/pmc_probe.py > /root/repo/gpurun_out/pmc2.log 2>&1
ls -R /root/repo/gpurun_out/pmc1 | head; tail -3 /root/repo/gpurun_out/pmc1.log
