#!/bin/bash
# The evidence of round 6, second call: counters + bench lines of the per-GPU shares of configurations 4 and 5 and of the rays aimed at the clustered scene's blobs; the
# lines without counters (aimed rays unbinned and at 4M, the soup at --snd-density 5, the whole batches of 4 and 5); construction timelines, profile and traffic of three
# scene families; the policy-regret table; the multi-GPU pre-flight.   usage: tools/gpu_round6b.sh TAG
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r6z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
export ESSENTIAL=1 PASS_LIMIT=120
tools/gpu_traffic_config.sh $TAG 4 --shard 3/8 > $OUT/traffic4.log 2>&1; cp $OUT/config4/traffic_config4.json profiles/ 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 2 --config 4 --shard 3/8 --no-cpu-baseline > $OUT/bench_config4_shard.json 2> $OUT/bench_config4_shard.err; cut -c1-160 $OUT/bench_config4_shard.json
PASS_LIMIT=200 tools/gpu_traffic_config.sh $TAG 5 --shard 3/8 > $OUT/traffic5.log 2>&1; cp $OUT/config5/traffic_config5.json profiles/ 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 2 --config 5 --shard 3/8 --no-cpu-baseline > $OUT/bench_config5_shard.json 2> $OUT/bench_config5_shard.err; cut -c1-160 $OUT/bench_config5_shard.json
TRAFFIC_SUFFIX=_aimed tools/gpu_traffic_config.sh $TAG 6 --rays aimed > $OUT/traffic6_aimed.log 2>&1; cp $OUT/config6_aimed/traffic_config6_aimed.json profiles/ 2>/dev/null
B="python bench.py --gpus 1 --no-cpu-baseline --inflight 0"
timeout 100 $B --steps 20 --warmup 3 --config clustered --rays aimed > $OUT/bench_clustered_aimed.json 2> $OUT/bench_clustered_aimed.err; cut -c1-160 $OUT/bench_clustered_aimed.json
timeout 100 $B --steps 20 --warmup 3 --config clustered --rays aimed --bin-rays 0 > $OUT/bench_clustered_aimed_unbinned.json 2> $OUT/bench_clustered_aimed_unbinned.err; cut -c1-160 $OUT/bench_clustered_aimed_unbinned.json
timeout 100 $B --steps 10 --warmup 3 --config clustered --rays aimed --total-rays 4194304 > $OUT/bench_clustered_aimed_4M.json 2> $OUT/bench_clustered_aimed_4M.err; cut -c1-160 $OUT/bench_clustered_aimed_4M.json
timeout 100 $B --steps 10 --warmup 2 --config 3 --snd-density 5.0 --top-density 0.12 > $OUT/bench_soup_sd5_4096.json 2> $OUT/bench_soup_sd5_4096.err; cut -c1-160 $OUT/bench_soup_sd5_4096.json
timeout 120 $B --steps 10 --warmup 2 --config 4 > $OUT/bench_config4.json 2> $OUT/bench_config4.err; cut -c1-160 $OUT/bench_config4.json
timeout 150 $B --steps 10 --warmup 2 --config 5 > $OUT/bench_config5.json 2> $OUT/bench_config5.err; cut -c1-160 $OUT/bench_config5.json
bash tools/gpu_build_timeline.sh $TAG soup clustered stadium 2>&1 | grep "build_ms_mean\|one construction\|kernels " | cut -c1-200
TRIS=8000000 ITERS=3 timeout 200 python tools/dev_build_time.py > $OUT/build_time_8M.txt 2>/dev/null; tail -1 $OUT/build_time_8M.txt | cut -c1-200
timeout 300 bash tools/gpu_build_traffic.sh $TAG > $OUT/build_traffic.log 2>&1; tail -5 $OUT/build_traffic.log | cut -c1-200
timeout 1500 python tools/dev_policy_regret.py > $OUT/policy_regret.txt 2> $OUT/policy_regret.err; sed -n '/| scene | batch/,$p' $OUT/policy_regret.txt | cut -c1-160
timeout 300 bash tools/scale_preflight.sh ${TAG}_preflight > $OUT/scale_preflight.txt 2>&1; tail -3 $OUT/scale_preflight.txt | cut -c1-160
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
