#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 67: the head share's threshold (tiles that cost N tenths of the median start with four lanes per ray, first; 20 by default)
OUT=gpurun_out/r6geo; mkdir -p $OUT
for sc in clustered stadium gradient shell ""; do
  for b in "primary 1024^2" "primary 1920x1080"; do
    echo "== ${sc:-soup} $b"
    SCENE=$sc timeout 300 python tools/dev_option_sweep.py traverse.quad_head 20,12,15,30,40,0 --batch "$b" --reps 1 --launches 200 2>&1 | grep "ms_median\|rror" | cut -c10-110
  done
done | tee $OUT/quad_head.txt
