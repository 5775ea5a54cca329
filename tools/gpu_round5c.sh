#!/bin/bash
# The evidence of round 5, last call (nine GPU minutes left): counters and bench lines at the final kernel sources, the most important first -- configuration 2, the clustered
# scene, configuration 3, the per-GPU shares of 4 and 5.  (The construction kernels and their evidence -- build_prof, construction traffic -- are those of the call before: only
# the host-side policy of traverse.hip changed since.)   usage: tools/gpu_round5c.sh TAG
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
python -c "from hagrid_amd import build as b; print('kernel sources', b.source_hash())" | tee $OUT/source_hash.txt
export ESSENTIAL=1 PASS_LIMIT=120
tools/gpu_traffic_config.sh $TAG 2 > $OUT/traffic2.log 2>&1; cp $OUT/config2/traffic_config2.json profiles/ 2>/dev/null
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench.json
tools/gpu_traffic_config.sh $TAG 6 > $OUT/traffic6.log 2>&1; cp $OUT/config6/traffic_config6.json profiles/ 2>/dev/null
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 3 --config clustered --no-cpu-baseline > $OUT/bench_clustered.json 2> $OUT/bench_clustered.err; cut -c1-200 $OUT/bench_clustered.json
tools/gpu_traffic_config.sh $TAG 3 > $OUT/traffic3.log 2>&1; cp $OUT/config3/traffic_config3.json profiles/ 2>/dev/null
timeout 200 python bench.py --gpus 1 --steps 10 --warmup 2 --config 3 --no-cpu-baseline > $OUT/bench_config3.json 2> $OUT/bench_config3.err; cut -c1-160 $OUT/bench_config3.json
tools/gpu_traffic_config.sh $TAG 4 --shard 3/8 > $OUT/traffic4.log 2>&1; cp $OUT/config4/traffic_config4.json profiles/ 2>/dev/null
timeout 200 python bench.py --gpus 1 --steps 10 --warmup 2 --config 4 --shard 3/8 --no-cpu-baseline > $OUT/bench_config4_shard.json 2> $OUT/bench_config4_shard.err; cut -c1-160 $OUT/bench_config4_shard.json
PASS_LIMIT=200 tools/gpu_traffic_config.sh $TAG 5 --shard 3/8 > $OUT/traffic5.log 2>&1; cp $OUT/config5/traffic_config5.json profiles/ 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 2 --config 5 --shard 3/8 --no-cpu-baseline > $OUT/bench_config5_shard.json 2> $OUT/bench_config5_shard.err; cut -c1-160 $OUT/bench_config5_shard.json
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
