#!/bin/bash
# 32-point query kernel: two workgroups per CU with deeper weight prefetch rings (side builds)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r04g}; mkdir -p $out
cd $R
{
echo "== 64-point kernel (product)"; MONOPORT_QUERY_SMALL_TILES=0 python tools/ablate.py run full f32 262144 1048576
echo "== 32-point kernel (product: 3 WG/CU, PF1 = 1, PF0 = 3)"; MONOPORT_QUERY_SMALL_TILES=1 python tools/ablate.py run full f32 262144 1048576
for v in a b c d e; do echo "== t32$v"; MONOPORT_QUERY_SMALL_TILES=1 python tools/ablate.py run t32$v f32 262144 1048576; done
} 2>&1 | grep -v amdgpu.ids | tee $out/t32_prefetch.log
