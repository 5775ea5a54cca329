#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 25: evidence repair -- the clustered scene's counters again (its directory was overwritten by the aimed batch's),
# the aimed batch's counters in a directory of their own and its bench lines with them.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=r6z; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
python -c "from hagrid_amd import build as b; print('kernel sources', b.source_hash())"
export ESSENTIAL=1 PASS_LIMIT=120
rm -rf $OUT/config6
tools/gpu_traffic_config.sh $TAG 6 > $OUT/traffic6.log 2>&1; cp $OUT/config6/traffic_config6.json profiles/ 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --config clustered --no-cpu-baseline > $OUT/bench_clustered.json 2> $OUT/bench_clustered.err; cut -c1-200 $OUT/bench_clustered.json
TRAFFIC_SUFFIX=_aimed tools/gpu_traffic_config.sh $TAG 6 --rays aimed > $OUT/traffic6_aimed.log 2>&1; cp $OUT/config6_aimed/traffic_config6_aimed.json profiles/ 2>/dev/null
B="python bench.py --gpus 1 --no-cpu-baseline --inflight 0"
timeout 100 $B --steps 20 --warmup 3 --config clustered --rays aimed > $OUT/bench_clustered_aimed.json 2> $OUT/bench_clustered_aimed.err; cut -c1-160 $OUT/bench_clustered_aimed.json
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
