#!/bin/bash
# One-off experiment round (round 4): the mailbox of the tail kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_traverse_gpu.py -x -q -k "tail_mode or every_traversal or binning" > $OUT/pytest_trav.log 2>&1; tail -4 $OUT/pytest_trav.log
timeout 900 python tools/dev_fuzz_kernels.py 8 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt
B="python bench.py --gpus 1 --steps 10 --warmup 3 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare"
for c in "4 --shard 3/8" "5 --shard 3/8" "3" "2"; do
  for v in 0 1; do
    n=$(echo "$c" | tr -c 'a-z0-9' '_')
    timeout 900 $B --config $c --opts traverse.mailbox=$v > $OUT/c${n}_mb$v.json 2> $OUT/c${n}_mb$v.err
    python - $OUT/c${n}_mb$v.json "config $c mailbox=$v" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value", j["value"], "ms_per_step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
  done
done
