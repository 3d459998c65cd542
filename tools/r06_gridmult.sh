cd $GRAFT_REPO_ROOT; out=gpurun_out/r06q; mkdir -p $out
for m in 1 2 4 8 16; do
  echo "=== MONOPORT_QUERY_GRID_MULT=$m" | tee -a $out/gridmult.txt
  MONOPORT_QUERY_GRID_MULT=$m timeout 300 python tools/single_frame_levels.py 1 4 20 2>&1 | grep -v amdgpu.ids | tee -a $out/gridmult.txt
done
