#!/bin/bash
# One-off experiment round (round 4): which resource binds the incoherent / beyond-cache configurations, binning variants, moving camera.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
B="python bench.py --gpus 1 --steps 5 --warmup 2 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare"
for v in "0" "1" "1 --opts traverse.bin_bits=3"; do
  n=$(echo "$v" | tr -c 'a-z0-9' '_')
  timeout 900 $B --config 5 --shard 3/8 --bin-rays $v > $OUT/c5_bin_$n.json 2> $OUT/c5_bin_$n.err; echo "c5 bin-rays $v: $(cut -c1-120 $OUT/c5_bin_$n.json)"
done
for v in "traverse.bin_bits=4" "traverse.lds_pad=6500" "traverse.lds_pad=9900" "traverse.tail_dual=1"; do
  timeout 900 $B --config 4 --shard 3/8 --opts $v > $OUT/c4_$v.json 2> $OUT/c4_$v.err; echo "c4 $v: $(cut -c1-120 $OUT/c4_$v.json)"
done
timeout 900 $B --config 5 --shard 3/8 --opts traverse.lds_pad=9900 > $OUT/c5_lds9900.json 2> $OUT/c5_lds9900.err; echo "c5 lds_pad 9900: $(cut -c1-120 $OUT/c5_lds9900.json)"
timeout 600 tools/dev_ab.sh "incoherent 4M binned" "primary 4096^2" > $OUT/ab_nops.txt 2>&1; cat $OUT/ab_nops.txt
timeout 900 python tools/dev_moving_camera.py > $OUT/moving_camera.txt 2>&1; cat $OUT/moving_camera.txt
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
