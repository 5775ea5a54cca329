#!/bin/bash
# One GPU-box round: build, GPU tests, smoke, bench, rocprofv3 kernel trace + PMC passes of the same bench command.
# usage: tools/gpu_round.sh TAG [notests]
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
if [ "$2" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
fi
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
export TMPDIR=/tmp
ROOT=$PWD
BENCH="python $ROOT/bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --inflight 0 --no-order-compare"   # (the two-in-flight extra stretches its launches: kept out of the per-kernel averages)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o trace -- $BENCH > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof.err)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_fetch -o pmc -- $BENCH > /dev/null 2> $ROOT/$OUT/pmc_fetch.err)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_write -o pmc -- $BENCH > /dev/null 2> $ROOT/$OUT/pmc_write.err)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $ROOT/$OUT/pmc_l2 -o pmc -- $BENCH > /dev/null 2> $ROOT/$OUT/pmc_l2.err)
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
# keep the merge small: drop raw traces, keep stats + summaries
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
