#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 59: bounce rays on a grid with two layouts after the choice by row kind
OUT=gpurun_out/r6two; mkdir -p $OUT
for wh in "2048 2048" "4096 4096"; do set -- $wh
for o in "" "traverse.image_uniform=0"; do
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 2 --config 3 --rays bounce --width $1 --height $2 --no-cpu-baseline --no-order-compare --inflight 1 --opts "$o" > $OUT/b.json 2> $OUT/b.err
  python - "$o" $1 <<'PY'
import json, sys
d = json.load(open('gpurun_out/r6two/b.json'))
print("bounce %s^2  opts %-28s %8.1f Mrays/s  %.4f ms  %s" % (sys.argv[2], sys.argv[1] or "(defaults)", d["value"], d["ms_per_step"], d["config"].get("traversal_image", "")[:60]))
PY
done; done | tee $OUT/bounce_layout2.txt
timeout 2700 python -m pytest tests/test_traverse_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-250
