#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 70: six candidates in the share trial (37 and 25 per cent added): tests, the default order and the moving camera on four scenes
OUT=gpurun_out/r6six; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_traverse_gpu.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
for sc in clustered stadium shell ""; do
  for b in "primary 1024^2" "primary 1920x1080"; do
    echo "== ${sc:-soup} $b (default order; the learned order)"
    SCENE=$sc timeout 300 python tools/dev_option_sweep.py traverse.tile_order 0,-1 --batch "$b" --reps 2 --launches 200 2>&1 | grep "ms_median\|rror" | cut -c10-110
  done
done | tee $OUT/default_order.txt
for sc in clustered stadium soup; do
  timeout 300 python tools/dev_frame_policies.py --scene $sc --frames 48 --sets "policy:;default_order:traverse.tile_order=0" 2>&1 | grep '"set"' | cut -c1-130
done | tee $OUT/frames.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --config clustered --no-cpu-baseline > $OUT/bench_clustered.json 2> $OUT/bench_clustered.err; cut -c1-200 $OUT/bench_clustered.json
