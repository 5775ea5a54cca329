#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 45: after the pruning (padded triangles, MOVING mode, the unmeasured limits): the GPU tests that touch it and the affected bench lines
OUT=gpurun_out/r6prune; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_traverse_gpu.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench.json
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 2 --config 4 --shard 3/8 --no-cpu-baseline > $OUT/bench_config4_shard.json 2> $OUT/bench_config4_shard.err; cut -c1-200 $OUT/bench_config4_shard.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --config clustered --no-cpu-baseline > $OUT/bench_clustered.json 2> $OUT/bench_clustered.err; cut -c1-200 $OUT/bench_clustered.json
