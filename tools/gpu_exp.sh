#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 65: super-tiles of 16 x 16 tiles and bands of 8 rows on the large launches (8 rounds and more)
OUT=gpurun_out/r6geo; mkdir -p $OUT
for o in "" "traverse.super_tile=4" "traverse.super_tile=4,traverse.band_rows=8" "traverse.super_tile=4,traverse.band_rows=2" "traverse.super_tile=5,traverse.band_rows=2"; do
  timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --config 5 --no-cpu-baseline --no-order-compare --inflight 1 --opts "$o" > $OUT/c5.json 2> $OUT/c5.err
  python - "$o" <<'PY'
import json, sys
d = json.load(open('gpurun_out/r6geo/c5.json'))
print("config 5 whole  opts %-44s %8.1f Mrays/s  %.4f ms" % (sys.argv[1] or "(defaults)", d["value"], d["ms_per_step"]))
PY
done | tee $OUT/large_geometry.txt
for sc in "" clustered stadium; do
  for b in "primary 4096^2" "primary 2048^2"; do
    echo "== ${sc:-soup} $b"
    SCENE=$sc timeout 300 python tools/dev_option_sweep.py traverse.super_tile 3,4 --batch "$b" --reps 2 --launches 30 2>&1 | grep "ms_median\|rror" | cut -c10-110
  done
done | tee -a $OUT/large_geometry.txt
echo "== config3 4096^2" | tee -a $OUT/large_geometry.txt
timeout 300 python tools/dev_option_sweep.py traverse.super_tile 3,4 --batch "config3 4096^2" --reps 2 --launches 30 2>&1 | grep "ms_median\|rror" | cut -c10-110 | tee -a $OUT/large_geometry.txt
