#!/bin/bash
# One-off experiment (round 5, job 14): the whole GPU suite after the second prune (119 kernels: scans in the look-back form only, any-hit as a run-time flag,
# image builders with depth and id width as arguments, one construction-format kernel per cell format); same-box A/B against the library before it (ab/libOLD.so).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log | cut -c1-300
cp hagrid_amd/libhagrid_amd.so /tmp/libNEW.so
for round in 1 2; do
for v in NEW OLD; do
  cp $( [ $v = NEW ] && echo /tmp/libNEW.so || echo ab/lib$v.so ) hagrid_amd/libhagrid_amd.so; touch hagrid_amd/libhagrid_amd.so
  echo "== $v (round $round)"
  timeout 600 python tools/dev_traverse_time.py 2>&1 | cut -c1-200
  [ $round = 1 ] && BATCH="primary 1024^2;incoherent 4M binned" FLAGS=1 timeout 600 python tools/dev_traverse_time.py 2>&1 | cut -c1-200
  [ $round = 1 ] && BATCH="primary 1024^2;incoherent 4M binned" FLAGS=2 timeout 600 python tools/dev_traverse_time.py 2>&1 | cut -c1-200
  [ $round = 1 ] && BATCH="primary 1024^2;incoherent 1M" OPTS=traverse.image=0 timeout 600 python tools/dev_traverse_time.py 2>&1 | cut -c1-200
  [ $round = 1 ] && timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --config clustered --no-cpu-baseline --inflight 0 --no-order-compare 2> $OUT/c.err | cut -c1-260
done
done
cp /tmp/libNEW.so hagrid_amd/libhagrid_amd.so
