#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r06v}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for t in 0x1000 0x800 0x400; do
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pa_$t -- python $R/tools/wino_pmc_probe.py $t > $out/pa.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $out/pb_$t -- python $R/tools/wino_pmc_probe.py $t > $out/pb.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM --output-format csv -d $out/pc_$t -- python $R/tools/wino_pmc_probe.py $t > $out/pc.log 2>&1
python - $out/pa_$t $out/pb_$t $out/pc_$t $t <<'PY'
import csv, glob, os, sys
from collections import defaultdict
for d in sys.argv[1:4]:
    rows = defaultdict(dict)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "conv3x3" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
                k = int(r["Dispatch_Id"]); rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                rows[k]["_name"] = r["Kernel_Name"][:40]
    if rows:
        last = sorted(rows)[-1]
        print(sys.argv[4], rows[last].pop("_name"), " ".join("%s=%.4g" % kv for kv in sorted(rows[last].items())))
PY
done 2>&1 | tee $out/wino_pmc.txt
rm -rf $out/pa_* $out/pb_* $out/pc_*
