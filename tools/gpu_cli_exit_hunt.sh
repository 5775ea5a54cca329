#!/bin/bash
# Hunt for the round-2 driver failure (hagrid_cli printed its report and never exited): the driver's own test sequence in its
# old (alphabetical) order, several times in fresh Python processes, then the bare command in a long loop.  A process that
# does not exit is diagnosed by tests/_subproc.py (wait channels, kernel and user stacks).
# usage: tools/gpu_cli_exit_hunt.sh TAG [sequence repeats] [loop rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-hunt}; REP=${2:-5}; ROUNDS=${3:-300}
OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
for i in $(seq 1 $REP); do
  HAGRID_TEST_ORDER=alpha timeout 600 python -m pytest tests/test_build_gpu.py tests/test_concurrency_gpu.py tests/test_cpp_api.py -m gpu -x -q -p no:cacheprovider > $OUT/seq_$i.log 2>&1
  echo "sequence $i: $(tail -1 $OUT/seq_$i.log)"
done
timeout 900 python tools/dev_cli_exit.py $ROUNDS --only-first > $OUT/loop.log 2>&1; tail -3 $OUT/loop.log
grep -l "TIMEOUT" $OUT/*.log
