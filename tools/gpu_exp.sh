#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 41: "traverse.step_cap" with more room in the queue
OUT=gpurun_out/r6cap; mkdir -p $OUT
for sc in clustered stadium; do
  for room in 25 50 100; do
    o="traverse.tile_order=0,traverse.share_trial=0,traverse.quad_tail=0,traverse.cap_room=$room"
    echo "== scene '${sc:-soup}' opts $o"
    SCENE=$sc OPTS=$o timeout 300 python tools/dev_option_sweep.py traverse.step_cap 0,32,48,64,96,128 --batch "primary 1024^2" --reps 1 --launches 100 2>&1 | grep "ms_median\|rror" | cut -c1-120
  done
done 2>&1 | tee $OUT/sweep_room.txt
