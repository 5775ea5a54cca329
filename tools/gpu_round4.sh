#!/bin/bash
# The evidence round of round 4: GPU tests, smoke, the bench lines of every configuration, kernel stats and counters per configuration
# (tools/gpu_traffic_config.sh), construction traffic, the multi-GPU pre-flight, the moving-camera loop.  usage: tools/gpu_round4.sh TAG [refresh]
# refresh: what depends on the kernel sources only (tests, counters, the lines that carry them); the whole-batch lines of configurations 4 and 5, the
# pre-flight, the fuzz and the moving-camera loop keep their last run.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r4z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 > $OUT/pytest_gpu.log 2>&1; tail -14 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
# counters first: the bench lines below carry them (profiles/traffic_config<C>.json must name these kernel sources)
tools/gpu_traffic_config.sh $TAG 2 > $OUT/traffic2.log 2>&1; cp $OUT/config2/traffic_config2.json profiles/ 2>/dev/null
tools/gpu_traffic_config.sh $TAG 3 > $OUT/traffic3.log 2>&1; cp $OUT/config3/traffic_config3.json profiles/ 2>/dev/null
tools/gpu_traffic_config.sh $TAG 4 --shard 3/8 > $OUT/traffic4.log 2>&1; cp $OUT/config4/traffic_config4.json profiles/ 2>/dev/null
tools/gpu_traffic_config.sh $TAG 5 --shard 3/8 > $OUT/traffic5.log 2>&1; cp $OUT/config5/traffic_config5.json profiles/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json; tail -2 $OUT/bench.err
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 --config 4 --shard 3/8 > $OUT/bench_config4_shard.json 2> $OUT/bench_config4_shard.err; cut -c1-200 $OUT/bench_config4_shard.json
timeout 1200 python bench.py --gpus 1 --steps 10 --warmup 2 --config 5 --shard 3/8 > $OUT/bench_config5_shard.json 2> $OUT/bench_config5_shard.err; cut -c1-200 $OUT/bench_config5_shard.json
for C in 3 $([ "$2" = refresh ] || echo 4 5); do
  timeout 1500 python bench.py --gpus 1 --steps 10 --warmup 2 --config $C > $OUT/bench_config$C.json 2> $OUT/bench_config$C.err
  echo "config $C rc=$?"; cut -c1-200 $OUT/bench_config$C.json
done
tools/gpu_build_traffic.sh ${TAG}_build > $OUT/build_traffic.log 2>&1; tail -3 $OUT/build_traffic.log
timeout 300 python tools/dev_merge_passes.py > $OUT/merge_passes.txt 2>&1; cut -c1-160 $OUT/merge_passes.txt
[ "$2" = refresh ] && { du -sh $OUT; exit 0; }
tools/scale_preflight.sh ${TAG}_preflight > $OUT/preflight.log 2>&1; tail -7 $OUT/preflight.log
timeout 900 python tools/dev_fuzz_kernels.py 20 > $OUT/fuzz.txt 2>&1; tail -1 $OUT/fuzz.txt
timeout 600 python tools/dev_moving_camera.py --speeds 0,0.25,1 > $OUT/moving_camera.txt 2>&1; tail -5 $OUT/moving_camera.txt
du -sh $OUT
