#!/bin/bash
# The evidence of round 5, the bench lines without counters once more at the final kernel sources (what is left of the GPU budget: five minutes): clustered scene with aimed rays,
# the soup at --snd-density 5, the whole batches of configurations 4 and 5, the multi-GPU pre-flight.   usage: tools/gpu_round5d.sh TAG
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
B="python bench.py --gpus 1 --no-cpu-baseline --inflight 0"
timeout 100 $B --steps 20 --warmup 3 --config clustered --rays aimed > $OUT/bench_clustered_aimed.json 2> $OUT/bench_clustered_aimed.err; cut -c1-160 $OUT/bench_clustered_aimed.json
timeout 100 $B --steps 10 --warmup 2 --config 3 --snd-density 5.0 --top-density 0.12 > $OUT/bench_soup_sd5_4096.json 2> $OUT/bench_soup_sd5_4096.err; cut -c1-160 $OUT/bench_soup_sd5_4096.json
timeout 120 $B --steps 10 --warmup 2 --config 4 > $OUT/bench_config4.json 2> $OUT/bench_config4.err; cut -c1-160 $OUT/bench_config4.json
timeout 150 $B --steps 10 --warmup 2 --config 5 > $OUT/bench_config5.json 2> $OUT/bench_config5.err; cut -c1-160 $OUT/bench_config5.json
timeout 200 bash tools/scale_preflight.sh ${TAG}_preflight > $OUT/scale_preflight.txt 2>&1; tail -3 $OUT/scale_preflight.txt | cut -c1-160
