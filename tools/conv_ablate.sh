#!/bin/bash
# Timing experiments on side builds of csrc/conv3x3.hip (never the product): what the activation
# reads, the output writes and the weight stream cost the convolution kernels at batch 10.
#   build (no GPU):  bash tools/conv_ablate.sh build        run (GPU box):  bash tools/conv_ablate.sh run
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "$1" = build ]; then
  for v in NOLOAD NOSTORE AHOT; do python tools/ablate.py build conv$v conv3x3.hip -DMPC_$v; done
  python tools/ablate.py build convNOIO conv3x3.hip -DMPC_NOLOAD -DMPC_NOSTORE
else
  for v in "" convNOLOAD convNOSTORE convAHOT convNOIO; do
    echo "== ${v:-product}"
    if [ -n "$v" ]; then export MONOPORT_HIP_LIB=$R/monoport_amd/lib/libmp_ablate$v.so; else unset MONOPORT_HIP_LIB; fi
    MODES=auto python tools/conv_bench.py ${BATCH:-10} 2>&1 | grep -v amdgpu.ids
  done
fi
