#!/bin/bash
# One-off experiment (round 5, job 2): the general layout of slim records (a record per voxel-map entry) -- parity tests, the clustered scene and a
# soup at --snd-density 5 against the 32-byte records and the construction format; construction time with the look-back scan's ack wait.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 2400 python -m pytest tests/test_traverse_gpu.py tests/test_build_gpu.py tests/test_scan_gpu.py -x -q > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log | cut -c1-300
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "clustered" > $OUT/pytest_clustered.log 2>&1; tail -5 $OUT/pytest_clustered.log | cut -c1-300
timeout 300 python tools/dev_build_time.py 2>&1 | tail -1 | cut -c1-400
timeout 600 python tools/dev_nonuniform.py frames > $OUT/nonuniform_frames.txt 2>&1; cut -c1-500 $OUT/nonuniform_frames.txt
timeout 600 python tools/dev_nonuniform.py compressed > $OUT/nonuniform_compressed.txt 2>&1; cut -c1-500 $OUT/nonuniform_compressed.txt
for slim in 1 0; do
OPTS=traverse.image_slim=$slim SD=5.0 timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0,1 --batch "primary 1024^2" --reps 1 2>&1 | cut -c1-300
OPTS=traverse.image_slim=$slim SD=5.0 timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0 --batch "primary 4096^2" --reps 1 --launches 20 2>&1 | cut -c1-300
done
timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0 --batch "config3 4096^2" --reps 1 --launches 20 2>&1 | cut -c1-300
B="python bench.py --gpus 1 --steps 10 --warmup 3 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare --hits-hash"
run() { timeout 900 $B $2 > $OUT/x.json 2> $OUT/x.err; python - $OUT/x.json "$1" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:60s} ms_per_step {j['ms_per_step']:8.4f}  Mrays/s {j['value']:8.1f}  hits {j['hits_sha256'][:12]}")
except Exception as e: print(sys.argv[2], "FAILED", e, open(sys.argv[1][:-5] + ".err").read()[-400:])
PY
}
run "config 5 share 3/8 default policy" "--config 5 --shard 3/8"
run "config 5 share 3/16 refill=0 (order rule)" "--config 5 --shard 3/16 --opts traverse.refill=0"
run "config 5 share 3/8 refill=0 (order rule)" "--config 5 --shard 3/8 --opts traverse.refill=0"
