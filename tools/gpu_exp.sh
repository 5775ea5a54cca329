#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 21: the gentler re-trial: policy tests, the full-size scenes, the trial's state over 1300 launches.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6t; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_traverse_gpu.py -m gpu -q -x -k "tile_order or head_share or share_trial or lifetime or tile_packets or binning" 2>&1 | tail -3 | cut -c1-300
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "stadium or clustered or config2_loop" 2>&1 | tail -3 | cut -c1-300
python tools/dev_order_state.py stadium 1024x1024 130 2>&1 | grep -v amdgpu | cut -c22-330 | awk 'NR%6==1' | cut -c1-60
