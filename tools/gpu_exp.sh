#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 66: the uniform layout's tail kernel WITHOUT the cost bookkeeping for launches that keep no costs (library B) against the one instantiation for all (A)
OUT=gpurun_out/r6geo; mkdir -p $OUT
cp hagrid_amd/libhagrid_amd.so /tmp/libA.so
for v in A B A B; do
  if [ $v = A ]; then cp /tmp/libA.so hagrid_amd/libhagrid_amd.so; else cp ab/libB.so hagrid_amd/libhagrid_amd.so; fi
  for cfg in "3" "4 --shard 3/8" "5 --shard 3/8" "5" "2 --width 4096 --height 4096"; do
    timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --config $cfg --no-cpu-baseline --no-order-compare --inflight 1 > $OUT/b.json 2> $OUT/b.err
    python - $v "$cfg" <<'PY'
import json, sys
d = json.load(open('gpurun_out/r6geo/b.json'))
print("lib %s  config %-28s %8.1f Mrays/s  %.4f ms" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"]))
PY
  done
done | tee $OUT/uniform_nocost.txt
cp /tmp/libA.so hagrid_amd/libhagrid_amd.so
