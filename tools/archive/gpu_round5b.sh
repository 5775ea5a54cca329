#!/bin/bash
# The evidence of round 5, second call: counters and bench line of the per-GPU share of configuration 5, the whole batches of configurations 4 and 5, the
# multi-GPU pre-flight on the one GPU.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
ESSENTIAL=1 tools/gpu_traffic_config.sh $TAG 5 --shard 3/8 > $OUT/traffic5.log 2>&1; cp $OUT/config5/traffic_config5.json profiles/ 2>/dev/null
timeout 1200 python bench.py --gpus 1 --steps 10 --warmup 2 --config 5 --shard 3/8 > $OUT/bench_config5_shard.json 2> $OUT/bench_config5_shard.err; cut -c1-160 $OUT/bench_config5_shard.json
for C in 4 5; do
  timeout 1500 python bench.py --gpus 1 --steps 10 --warmup 2 --config $C > $OUT/bench_config$C.json 2> $OUT/bench_config$C.err
  echo "config $C rc=$?"; cut -c1-160 $OUT/bench_config$C.json
done
timeout 1500 bash tools/scale_preflight.sh ${TAG}_preflight > $OUT/scale_preflight.txt 2>&1; tail -12 $OUT/scale_preflight.txt | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
