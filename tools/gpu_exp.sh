#!/bin/bash
# One-off experiment (round 5, job 1): the refill path merged -- traversal tests, configuration 5 / 4 shares with the default policy, the tile order by launch
# length on primary and on bounce rays, the clustered scene and a shift-4 soup as they are before the general slim image.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 1800 python -m pytest tests/test_traverse_gpu.py tests/test_build_gpu.py tests/test_scan_gpu.py -x -q > $OUT/pytest_traverse.log 2>&1; tail -3 $OUT/pytest_traverse.log
timeout 300 python tools/dev_build_time.py 2>&1 | tail -4
B="python bench.py --gpus 1 --steps 10 --warmup 3 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare --hits-hash"
run() { timeout 900 $B $2 > $OUT/x.json 2> $OUT/x.err; python - $OUT/x.json "$1" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:60s} ms_per_step {j['ms_per_step']:8.4f}  Mrays/s {j['value']:8.1f}  hits {j['hits_sha256'][:12]}")
except Exception as e: print(sys.argv[2], "FAILED", e, open(sys.argv[1][:-5] + ".err").read()[-400:])
PY
}
for sh in 3/8 3/16 3/32 3/64; do
  run "config 5 share $sh default policy" "--config 5 --shard $sh"
  run "config 5 share $sh refill=0 tile_order=0" "--config 5 --shard $sh --opts traverse.refill=0,traverse.tile_order=0"
  run "config 5 share $sh refill=0 tile_order=1" "--config 5 --shard $sh --opts traverse.refill=0,traverse.tile_order=1"
  run "config 5 share $sh refill=2" "--config 5 --shard $sh --opts traverse.refill=2"
done
run "config 5 whole batch default policy" "--config 5"
run "config 4 share 3/8 default policy" "--config 4 --shard 3/8"
for b in "primary 1536x1536" "primary 2048x2048" "primary 2560x2560" "primary 3072x3072"; do
  timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0,1 --batch "$b" --reps 2 --launches 50 2>&1 | grep -v '"grid"' | cut -c1-200
done
timeout 600 python tools/dev_nonuniform.py frames > $OUT/nonuniform_frames.txt 2>&1; cut -c1-400 $OUT/nonuniform_frames.txt
SD=5.0 timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0,1 --batch "primary 1024^2" --reps 1 2>&1 | cut -c1-300
SD=5.0 timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0 --batch "primary 4096^2" --reps 1 --launches 20 2>&1 | cut -c1-300
