#!/bin/bash
# bench.py lines of the BASELINE configurations 3, 4 and 5 on one GPU (the whole batch of each); usage: tools/gpu_configs.sh TAG
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-cfg}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
for C in 3 4 5; do
  timeout 1500 python bench.py --gpus 1 --steps 10 --warmup 2 --config $C > $OUT/bench_config$C.json 2> $OUT/bench_config$C.err
  echo "config $C rc=$?"; cut -c1-600 $OUT/bench_config$C.json; tail -2 $OUT/bench_config$C.err
done
