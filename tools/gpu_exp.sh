#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 29: confirmation samples of the default order next to the order's own: configuration 3 with the full bench line, policy tests, large cells.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6x; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_traverse_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "tile_order or head_share or share_trial or config3 or config2_loop or clustered" 2>&1 | tail -3 | cut -c1-300
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 2 --config 3 --no-cpu-baseline > $OUT/bench_config3.json 2> $OUT/bench_config3.err; cut -c1-160 $OUT/bench_config3.json
python tools/dev_order_state.py soup 4096x4096 12 2>&1 | grep -v amdgpu | cut -c1-330 | awk 'NR%2==0' | cut -c1-40,100-330
timeout 1500 python tools/dev_policy_regret.py --scenes soup,stadium,gradient --sizes 4096x4096,1024x1024 --kinds primary > $OUT/policy_regret.txt 2> $OUT/policy_regret.err; sed -n '/| scene | batch/,$p' $OUT/policy_regret.txt | grep "^|" | cut -c1-200
