#!/bin/bash
# One-off experiment round (round 4): padded triangle copy, merge iterations in place.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 1200 python -m pytest tests/test_build_gpu.py tests/test_scan_gpu.py -x -q --durations=5 > $OUT/pytest_build.log 2>&1; tail -8 $OUT/pytest_build.log
for o in "merge.inplace=1" "merge.inplace=0"; do echo "== $o"; OPTS=$o ITERS=10 timeout 300 python tools/dev_build_time.py 2>&1 | tail -1; done
TRIS=8000000 ITERS=3 timeout 600 python tools/dev_build_time.py 2>&1 | tail -1
TRIS=8000000 ITERS=3 OPTS=merge.inplace=0 timeout 600 python tools/dev_build_time.py 2>&1 | tail -1
B="python bench.py --gpus 1 --steps 10 --warmup 3 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare"
for c in "4 --shard 3/8" "5 --shard 3/8" "3" "2"; do
  for v in 1 0; do
    n=$(echo "$c" | tr -c 'a-z0-9' '_')
    timeout 900 $B --config $c --opts traverse.tri_pad=$v > $OUT/c${n}_pad$v.json 2> $OUT/c${n}_pad$v.err
    python - $OUT/c${n}_pad$v.json "config $c tri_pad=$v" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value", j["value"], "ms_per_step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
  done
done
timeout 1500 python -m pytest tests/test_traverse_gpu.py tests/test_fullsize_gpu.py -x -q --durations=5 > $OUT/pytest_trav.log 2>&1; tail -8 $OUT/pytest_trav.log
