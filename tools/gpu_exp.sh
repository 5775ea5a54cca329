#!/bin/bash
# One-off experiment (round 5, job 9): the table layout with wide records (a soup at --snd-density 5), traversal tests, construction time after the merge revert.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 2400 python -m pytest tests/test_traverse_gpu.py tests/test_build_gpu.py -x -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log | cut -c1-400
timeout 300 python tools/dev_build_time.py 2>&1 | tail -1 | cut -c1-300
SD=5.0 timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0,1 --batch "primary 1024^2" --reps 1 2>&1 | cut -c1-300
SD=5.0 timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0 --batch "primary 4096^2" --reps 1 --launches 20 2>&1 | tail -1 | cut -c1-300
OPTS=traverse.image_general=2 SD=5.0 timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0 --batch "primary 4096^2" --reps 1 --launches 20 2>&1 | tail -1 | cut -c1-300
timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0 --batch "config3 4096^2" --reps 1 --launches 20 2>&1 | tail -1 | cut -c1-300
