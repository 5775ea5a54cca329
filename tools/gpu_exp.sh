#!/bin/bash
# One-off experiment round (round 4): the mailbox of the tail kernel; what the order gate and the bands cost the 1024^2 launch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
cat /sys/fs/cgroup/cpu.max; nproc; lscpu | grep -E "Socket|Core|Thread|NUMA node\(s\)|Model name"
timeout 900 python -m pytest tests/test_traverse_gpu.py -x -q -k "tail_mode or every_traversal or binning or tile" > $OUT/pytest_trav.log 2>&1; tail -4 $OUT/pytest_trav.log
timeout 900 python tools/dev_fuzz_kernels.py 8 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt
B="python bench.py --gpus 1 --steps 20 --warmup 3 --build-iter 1 --no-cpu-baseline --inflight 0"
run() { n=$(echo "$1" | tr -c 'a-z0-9' '_'); timeout 900 $B $1 > $OUT/$n.json 2> $OUT/$n.err; python - $OUT/$n.json "$1" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); t = j.get("tile_order") or {}
    print(sys.argv[2], "| value", j["value"], "ms_per_step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"], "default_order", t.get("ms_per_step_default_order"))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
run "--config 2"
run "--config 2 --opts traverse.order_gate=0"
run "--config 2 --opts traverse.band_rows=4"
run "--config 2 --opts traverse.band_rows=1"
done
for c in "4 --shard 3/8" "5 --shard 3/8" "3"; do
  for v in 0 1; do run "--config $c --no-order-compare --opts traverse.mailbox=$v"; done
done
run "--config 2 --opts traverse.mailbox=1"
