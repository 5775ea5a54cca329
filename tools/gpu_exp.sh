#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 49: after the second pruning (REFILL gone, the mailbox for binned batches only): traversal tests, configuration 5 (share and whole batch), the headline
OUT=gpurun_out/r6prune2; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_traverse_gpu.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench.json
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 2 --config 5 --shard 3/8 --no-cpu-baseline > $OUT/bench_config5_shard.json 2> $OUT/bench_config5_shard.err; cut -c1-200 $OUT/bench_config5_shard.json
timeout 300 python bench.py --gpus 1 --steps 5 --warmup 1 --config 5 --no-cpu-baseline > $OUT/bench_config5.json 2> $OUT/bench_config5.err; cut -c1-200 $OUT/bench_config5.json
