#!/bin/bash
# One-off experiment round (round 4): merge in place (late iterations), banded tile order.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 1200 python -m pytest tests/test_build_gpu.py -x -q > $OUT/pytest_build.log 2>&1; tail -4 $OUT/pytest_build.log
for o in "merge.inplace=1" "merge.inplace=0" "merge.inplace_div=10"; do echo "== $o"; OPTS=$o ITERS=10 timeout 300 python tools/dev_build_time.py 2>&1 | tail -1 | cut -c1-200; done
TRIS=8000000 ITERS=3 timeout 600 python tools/dev_build_time.py 2>&1 | tail -1 | cut -c1-200
ITERS=3 PYTHONPATH=$PWD tools/gpu_prof_cmd.sh ${TAG}_prof python $PWD/tools/dev_build_time.py | grep -E "ip_|merge|remap|widen|cell_flags" 
B="python bench.py --gpus 1 --steps 10 --warmup 3 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare"
for c in "5 --shard 3/8" "3" "2" "2 --width 2048 --height 2048" "4 --shard 3/8"; do
  for v in 1 0 4 22; do
    n=$(echo "$c" | tr -c 'a-z0-9' '_')
    [ "$c" = "4 --shard 3/8" ] && [ $v != 0 ] && continue
    timeout 900 $B --config $c --opts traverse.band_rows=$v > $OUT/c${n}_band$v.json 2> $OUT/c${n}_band$v.err
    python - $OUT/c${n}_band$v.json "config $c band_rows=$v" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value", j["value"], "ms_per_step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
  done
done
timeout 1500 python -m pytest tests/test_traverse_gpu.py -x -q > $OUT/pytest_trav.log 2>&1; tail -4 $OUT/pytest_trav.log
