#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r02s}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for prec in f16x3 f32; do
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pa_$prec -- python $R/tools/conv_pmc_probe.py $prec > $out/pa.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $out/pb_$prec -- python $R/tools/conv_pmc_probe.py $prec > $out/pb.log 2>&1
python - $out/pa_$prec $out/pb_$prec $prec <<'PY'
import csv, glob, os, sys
from collections import defaultdict
for d in sys.argv[1:3]:
    rows = defaultdict(dict)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "conv3x3_gn" in r["Kernel_Name"]:
                k = int(r["Dispatch_Id"]); rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    last = sorted(rows)[-1]
    print(sys.argv[3], " ".join("%s=%.4g" % kv for kv in sorted(rows[last].items())))
PY
done
rm -rf $out/pa_* $out/pb_*
