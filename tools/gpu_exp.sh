#!/bin/bash
# One-off experiment round (round 4): the mailbox with the tests of the four-lanes-per-ray phase recorded as well.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_traverse_gpu.py -x -q -k "tail_mode or every_traversal or binning or tile" > $OUT/pytest_trav.log 2>&1; tail -4 $OUT/pytest_trav.log
timeout 900 python tools/dev_fuzz_kernels.py 12 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt
B="python bench.py --gpus 1 --steps 20 --warmup 3 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare"
run() { n=$(echo "$1" | tr -c 'a-z0-9' '_'); timeout 900 $B $1 > $OUT/$n.json 2> $OUT/$n.err; python - $OUT/$n.json "$1" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "| value", j["value"], "ms_per_step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for c in "5 --shard 3/8" "4 --shard 3/8"; do for v in 0 1 -1; do run "--config $c --opts traverse.mailbox=$v"; done; done
run "--config 5"
run "--config 5 --opts traverse.mailbox=0"
