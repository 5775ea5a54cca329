#!/bin/bash
# One-off experiment (round 5, job 11): the table layout with / without wide records as two instantiations; configuration 3 against round 4's library (B).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 2400 python -m pytest tests/test_traverse_gpu.py -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log | cut -c1-400
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
SD=5.0 timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0 --batch "primary 4096^2" --reps 1 --launches 20 2>&1 | tail -1 | cut -c1-300
