#!/bin/bash
# One-off experiment (round 5, job 6): three slim layouts (uniform, table restored, general) -- traversal tests, configuration 3's grid against round 4's
# library; the in-place merge on lists, kernel by kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 2400 python -m pytest tests/test_traverse_gpu.py tests/test_build_gpu.py -x -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log | cut -c1-400
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "clustered" > $OUT/pytest_clustered.log 2>&1; tail -3 $OUT/pytest_clustered.log | cut -c1-300
timeout 300 python tools/dev_build_time.py 2>&1 | tail -1 | cut -c1-300
PYTHONPATH=$PWD bash tools/gpu_prof_cmd.sh $TAG/buildprof python $PWD/tools/dev_build_time.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-160 | head -40
cp hagrid_amd/libhagrid_amd.so /tmp/libA.so
ab() {  # batch, env
  for round in 1 2; do for v in A B; do
    cp $( [ $v = A ] && echo /tmp/libA.so || echo ab/lib$v.so ) hagrid_amd/libhagrid_amd.so; touch hagrid_amd/libhagrid_amd.so
    echo -n "$v $2 | "; env $2 python tools/dev_option_sweep.py traverse.tail 1 --reps 1 --launches ${3:-100} --batch "$1" 2>&1 | tail -1 | cut -c1-200
  done; done
  cp /tmp/libA.so hagrid_amd/libhagrid_amd.so
}
ab "config3 4096^2" "X=1" 20
ab "primary 1024^2" "TD=0.15 SD=3.0"
cp /tmp/libA.so hagrid_amd/libhagrid_amd.so
