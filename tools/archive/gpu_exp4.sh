cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r4w4; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
B="python $PWD/bench.py --gpus 1 --steps 10 --warmup 2 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare --config 5 --shard 3/8"
for o in "traverse.mailbox=1" "traverse.mailbox=0,traverse.tail_dual=0" "traverse.mailbox=0,traverse.tail_dual=1" "traverse.mailbox=0,traverse.tail_dual=1,traverse.tri_pad=1"; do
  timeout 600 $B --opts $o > $OUT/x.json 2> $OUT/x.err
  python - "$o" $OUT/x.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[2])); print(f"{sys.argv[1]:70s} ms_per_step {j['ms_per_step']}  kernel_ms {j['roofline']['kernel_ms']}  Mrays/s {j['value']}")
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
