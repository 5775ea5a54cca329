#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 24: later share trials sample the shortlisted candidates only: tests; the soup over 1300 launches.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_traverse_gpu.py -m gpu -q -x -k "tile_order or head_share or share_trial or lifetime or tile_packets or binning" 2>&1 | tail -3 | cut -c1-300
python tools/dev_order_state.py soup 1024x1024 125 2>&1 | grep -v amdgpu | cut -c1-60 | awk 'NR<4 || (NR>98 && NR<112)'
