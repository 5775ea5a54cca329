cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r4y2; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench.json; tail -1 $OUT/bench.err
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 2 --config 4 --shard 3/8 > $OUT/bench_config4_shard.json 2> $OUT/bench_config4_shard.err; cut -c1-160 $OUT/bench_config4_shard.json
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 2 --config 3 > $OUT/bench_config3.json 2> $OUT/bench_config3.err; cut -c1-160 $OUT/bench_config3.json
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 2 --config 5 --shard 3/8 > $OUT/bench_config5_shard.json 2> $OUT/bench_config5_shard.err; cut -c1-160 $OUT/bench_config5_shard.json
timeout 600 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -2
