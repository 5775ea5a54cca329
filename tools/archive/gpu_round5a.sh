#!/bin/bash
# The evidence of round 5, first call (a call may run for an hour): GPU tests, smoke, counters and bench lines of configurations 2, 3, the per-GPU share of 4 and
# the clustered scene (--config clustered = 6), construction per kernel.  usage: tools/gpu_round5a.sh TAG
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
python -c "from hagrid_amd import build as b; print('kernel sources', b.source_hash())" | tee $OUT/source_hash.txt
timeout 1800 python -m pytest tests -m gpu -x -q --durations=10 > $OUT/pytest_gpu.log 2>&1; tail -14 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
tools/gpu_traffic_config.sh $TAG 2 > $OUT/traffic2.log 2>&1; cp $OUT/config2/traffic_config2.json profiles/ 2>/dev/null
tools/gpu_traffic_config.sh $TAG 6 > $OUT/traffic6.log 2>&1; cp $OUT/config6/traffic_config6.json profiles/ 2>/dev/null
ESSENTIAL=1 tools/gpu_traffic_config.sh $TAG 4 --shard 3/8 > $OUT/traffic4.log 2>&1; cp $OUT/config4/traffic_config4.json profiles/ 2>/dev/null
ESSENTIAL=1 tools/gpu_traffic_config.sh $TAG 3 > $OUT/traffic3.log 2>&1; cp $OUT/config3/traffic_config3.json profiles/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json; tail -2 $OUT/bench.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 --config clustered > $OUT/bench_clustered.json 2> $OUT/bench_clustered.err; cut -c1-200 $OUT/bench_clustered.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 --config clustered --rays aimed --no-cpu-baseline --inflight 0 > $OUT/bench_clustered_aimed.json 2> $OUT/bench_clustered_aimed.err; cut -c1-200 $OUT/bench_clustered_aimed.json
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 --config 4 --shard 3/8 > $OUT/bench_config4_shard.json 2> $OUT/bench_config4_shard.err; cut -c1-160 $OUT/bench_config4_shard.json
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 --config 3 > $OUT/bench_config3.json 2> $OUT/bench_config3.err; cut -c1-160 $OUT/bench_config3.json
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 --config 3 --snd-density 5.0 --top-density 0.12 --no-cpu-baseline --inflight 0 > $OUT/bench_soup_sd5_4096.json 2> $OUT/bench_soup_sd5_4096.err; cut -c1-160 $OUT/bench_soup_sd5_4096.json
ITERS=5 PYTHONPATH=$PWD tools/gpu_prof_cmd.sh ${TAG}_buildprof python $PWD/tools/dev_build_time.py > $OUT/build_prof.txt 2>&1; head -30 $OUT/build_prof.txt | cut -c1-170
timeout 300 python tools/dev_build_time.py 2>&1 | tail -1 > $OUT/build_time.txt; cut -c1-300 $OUT/build_time.txt
TRIS=8000000 ITERS=3 timeout 600 python tools/dev_build_time.py 2>&1 | tail -1 > $OUT/build_time_8M.txt; cut -c1-300 $OUT/build_time_8M.txt
bash tools/gpu_build_traffic.sh ${TAG}_buildtraffic > /dev/null 2>&1; cp gpurun_out/${TAG}_buildtraffic/construction_traffic.txt $OUT/ 2>/dev/null; tail -3 $OUT/construction_traffic.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
