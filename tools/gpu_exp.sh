#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 22: the all-tiles candidate with an order of its own: tests, regret cells, state traces.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6u; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_traverse_gpu.py -m gpu -q -x -k "tile_order or head_share or share_trial or lifetime or tile_packets or binning" 2>&1 | tail -3 | cut -c1-300
python tools/dev_order_state.py clustered 1280x720 14 2>&1 | grep -v amdgpu | cut -c22-330 | cut -c1-12,40-300
python tools/dev_order_state.py stadium 1920x1080 18 2>&1 | grep -v amdgpu | cut -c22-330 | cut -c1-12,40-300
timeout 1500 python tools/dev_policy_regret.py --scenes clustered,stadium,soup,shell --sizes 640x480,1280x720,1024x1024,1920x1080 --kinds primary > $OUT/policy_regret.txt 2> $OUT/policy_regret.err; sed -n '/| scene | batch/,$p' $OUT/policy_regret.txt | grep "^|" | cut -c1-200
