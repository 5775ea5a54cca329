#!/bin/bash
# One-off experiment (round 5, job 7): the in-place merge on lists with batched atomics and full-occupancy sweeps, kernel by kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 2400 python -m pytest tests/test_build_gpu.py -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log | cut -c1-400
timeout 300 python tools/dev_build_time.py 2>&1 | tail -1 | cut -c1-300
OPTS=merge.inplace=0 timeout 300 python tools/dev_build_time.py 2>&1 | tail -1 | cut -c1-300
PYTHONPATH=$PWD bash tools/gpu_prof_cmd.sh $TAG/buildprof python $PWD/tools/dev_build_time.py 2>&1 | grep -v "amdgpu.ids" | grep "ip_\|merge\|remap" | cut -c1-160 | head -40
