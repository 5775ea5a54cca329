#!/bin/bash
# One-off experiment (round 5, job 22): construction tests and times at the chosen tile sizes, shift read back with the first level's cells; per-kernel profile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 1500 python -m pytest tests/test_build_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "build or structure or sizes or clustered or config2" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log | cut -c1-300
for r in 1 2; do ITERS=10 timeout 300 python tools/dev_build_pool.py 2>&1 | tail -1 | cut -c1-400; done
ITERS=5 PYTHONPATH=$PWD tools/gpu_prof_cmd.sh ${TAG}_prof python $PWD/tools/dev_build_time.py 2>&1 | head -3 | cut -c1-170
