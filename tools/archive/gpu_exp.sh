#!/bin/bash
# One-off experiment (round 5, job 31): the new parity test of the head share on two more scene families.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.build()" || exit 1
HAGRID_TRACE_HEAD=1 timeout 150 python -m pytest tests/test_traverse_gpu.py -m gpu -q -x -s -k "head_share_trial" 2>&1 | grep -v "amdgpu" | tail -30 | cut -c1-200
