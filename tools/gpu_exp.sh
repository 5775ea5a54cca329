#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 33: one row of super-tiles per band as a candidate of the share trial; a concluded comparison stands across a relearn.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6y2; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_traverse_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "tile_order or head_share or share_trial or tile_packets or config3 or config5" 2>&1 | tail -3 | cut -c1-300
timeout 1500 python tools/dev_policy_regret.py --scenes stadium,soup,gradient --sizes 2048x2048,4096x4096 --kinds primary,bounce > $OUT/policy_regret.txt 2> $OUT/policy_regret.err; sed -n '/| scene | batch/,$p' $OUT/policy_regret.txt | grep "^|" | cut -c1-200
timeout 300 python tools/dev_moving_camera.py --scene soup --frames 40 --speeds 1 2>&1 | grep -v amdgpu | cut -c1-330
B="python bench.py --gpus 1 --no-cpu-baseline --inflight 0 --no-order-compare"
timeout 200 $B --steps 10 --warmup 2 --config 5 --shard 3/8 > $OUT/bench_config5_shard.json 2> $OUT/bench_config5_shard.err; cut -c1-160 $OUT/bench_config5_shard.json
timeout 200 $B --steps 10 --warmup 2 --config 3 > $OUT/bench_config3.json 2> $OUT/bench_config3.err; cut -c1-160 $OUT/bench_config3.json
