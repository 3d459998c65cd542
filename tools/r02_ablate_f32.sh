#!/bin/bash
# Round-2 A/B of the f32 query kernel's weight stream (run on the GPU box through gpurun).
# flat = round-1 kernel (64-bit flat pointers), full = product (buffer-resource loads),
# pfXY = product with deeper A-fragment prefetch rings.
out=gpurun_out/r02b; mkdir -p $out
sizes="19652 36254 100450 203685 847114 1048576"
for rep in 1 2; do
  for v in full flat pf12 pf1205 pf07 pf13; do
    python tools/ablate.py run $v f32 $sizes >> $out/ablate_f32.log 2>&1
  done
done
cat $out/ablate_f32.log | grep TFLOP
