#!/bin/bash
# One-off experiment (round 4): construction with this library against the one before the change (ab/libHEADgate.so), same box, alternating; per-kernel table of both.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_build_gpu.py -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for rep in 1 2 3; do
  for v in HEAD ${BASES:-HEADgate}; do
    if [ $v = HEAD ]; then unset HAGRID_AMD_LIB; else export HAGRID_AMD_LIB=$PWD/ab/lib$v.so; fi
    echo -n "$v rep $rep: "; ITERS=${ITERS:-15} timeout 300 python tools/dev_build_time.py 2>/dev/null | cut -c1-230
  done
done
for v in HEAD ${BASES:-HEADgate}; do
  if [ $v = HEAD ]; then unset HAGRID_AMD_LIB; else export HAGRID_AMD_LIB=$PWD/ab/lib$v.so; fi
  echo "== per kernel, $v"; ITERS=5 PYTHONPATH=$PWD tools/gpu_prof_cmd.sh ${TAG}_prof_$v python $PWD/tools/dev_build_time.py | head -${ROWS:-32} | cut -c1-150
done
