#!/bin/bash
# One-off experiment (round 5, job 27): the whole GPU suite with "traverse.quad_head" = 20 as the default; bench lines of configuration 2 and the clustered scene.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log | cut -c1-300
for c in "" "--config clustered"; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --inflight 0 $c 2> $OUT/b.err | cut -c1-220; tail -1 $OUT/b.err | cut -c1-200; done
