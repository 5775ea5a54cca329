#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 51: the construction's own switches on three scene families (are rounds 2 - 4's settings still the best?)
OUT=gpurun_out/r6build; mkdir -p $OUT
for sc in "" clustered stadium; do
  for o in "" "merge.inplace=0" "merge.inplace_div=3" "merge.inplace_div=4" "merge.inplace_div=8" "merge.inplace_div=1" "merge.narrow_cells=0" "expand.voxel_map=0" "scan.lookback=2"; do
    r=$(SCENE=$sc OPTS=$o ITERS=8 timeout 300 python tools/dev_build_time.py 2>&1 | grep build_ms_mean | cut -c1-200)
    echo "scene ${sc:-soup}  opts ${o:-(defaults)}  $r"
  done
done | tee $OUT/options.txt
