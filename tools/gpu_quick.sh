#!/bin/bash
# A short GPU-box round: build, the named test files (default: all GPU tests), one bench line.
# usage: tools/gpu_quick.sh TAG [pytest args...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-q}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
ARGS=${@:-tests -m gpu}
timeout 2400 python -m pytest $ARGS -x -q --durations=15 > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
