#!/bin/bash
# The evidence of round 4, first call (a call may run for an hour): GPU tests, smoke, counters and bench lines of configurations 2, 3 and the
# per-GPU share of 4, merge pass counts, construction traffic.  usage: tools/gpu_round4a.sh TAG
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r4z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 > $OUT/pytest_gpu.log 2>&1; tail -14 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
tools/gpu_traffic_config.sh $TAG 2 > $OUT/traffic2.log 2>&1; cp $OUT/config2/traffic_config2.json profiles/ 2>/dev/null
ESSENTIAL=1 tools/gpu_traffic_config.sh $TAG 4 --shard 3/8 > $OUT/traffic4.log 2>&1; cp $OUT/config4/traffic_config4.json profiles/ 2>/dev/null
ESSENTIAL=1 tools/gpu_traffic_config.sh $TAG 3 > $OUT/traffic3.log 2>&1; cp $OUT/config3/traffic_config3.json profiles/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json; tail -2 $OUT/bench.err
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 --config 4 --shard 3/8 > $OUT/bench_config4_shard.json 2> $OUT/bench_config4_shard.err; cut -c1-160 $OUT/bench_config4_shard.json
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 --config 3 > $OUT/bench_config3.json 2> $OUT/bench_config3.err; cut -c1-160 $OUT/bench_config3.json
timeout 300 python tools/dev_merge_passes.py > $OUT/merge_passes.txt 2>&1; cut -c1-200 $OUT/merge_passes.txt
tools/gpu_build_traffic.sh ${TAG}_build > $OUT/build_traffic.log 2>&1; tail -2 $OUT/build_traffic.log
ITERS=5 PYTHONPATH=$PWD tools/gpu_prof_cmd.sh ${TAG}_buildprof python $PWD/tools/dev_build_time.py > $OUT/build_prof.txt 2>&1; head -30 $OUT/build_prof.txt | cut -c1-170
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
