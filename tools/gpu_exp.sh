#!/bin/bash
# One-off experiment round (round 4): merge in place from stamps.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 1200 python -m pytest tests/test_build_gpu.py -x -q > $OUT/pytest_build.log 2>&1; tail -12 $OUT/pytest_build.log
for o in "merge.inplace=1" "merge.inplace=0" "merge.inplace_div=1"; do echo "== $o"; OPTS=$o ITERS=10 timeout 300 python tools/dev_build_time.py 2>&1 | tail -1 | cut -c1-200; done
TRIS=8000000 ITERS=3 timeout 600 python tools/dev_build_time.py 2>&1 | tail -1 | cut -c1-200
ITERS=3 PYTHONPATH=$PWD tools/gpu_prof_cmd.sh ${TAG}_prof python $PWD/tools/dev_build_time.py | grep -E "ip_|merge|remap|widen|cell_flags|Sums" 
timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -k "config2 or clustered" > $OUT/pytest_full.log 2>&1; tail -3 $OUT/pytest_full.log
