#!/bin/bash
# The evidence of round 6, first call: the GPU tests and smoke, then counters + bench lines at these kernel sources for configuration 2, the clustered scene, the stadium
# scene (config 7) and configuration 3.   usage: tools/gpu_round6a.sh TAG      (second call: tools/gpu_round6b.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r6z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
python -c "from hagrid_amd import build as b; print('kernel sources', b.source_hash())" | tee $OUT/source_hash.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | cut -c1-300 | tee $OUT/pytest_gpu_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log | cut -c1-200
export ESSENTIAL=1 PASS_LIMIT=120
tools/gpu_traffic_config.sh $TAG 2 > $OUT/traffic2.log 2>&1; cp $OUT/config2/traffic_config2.json profiles/ 2>/dev/null
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench_time.txt; cut -c1-200 $OUT/bench.json; grep real $OUT/bench_time.txt
tools/gpu_traffic_config.sh $TAG 6 > $OUT/traffic6.log 2>&1; cp $OUT/config6/traffic_config6.json profiles/ 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --config clustered --no-cpu-baseline > $OUT/bench_clustered.json 2> $OUT/bench_clustered.err; cut -c1-200 $OUT/bench_clustered.json
tools/gpu_traffic_config.sh $TAG 7 > $OUT/traffic7.log 2>&1; cp $OUT/config7/traffic_config7.json profiles/ 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --config stadium --no-cpu-baseline > $OUT/bench_stadium.json 2> $OUT/bench_stadium.err; cut -c1-200 $OUT/bench_stadium.json
tools/gpu_traffic_config.sh $TAG 3 > $OUT/traffic3.log 2>&1; cp $OUT/config3/traffic_config3.json profiles/ 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 2 --config 3 --no-cpu-baseline > $OUT/bench_config3.json 2> $OUT/bench_config3.err; cut -c1-160 $OUT/bench_config3.json
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
