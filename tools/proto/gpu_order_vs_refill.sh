cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/proto5; mkdir -p $OUT
B="python bench.py --gpus 1 --steps 10 --warmup 3 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare --hits-hash --config 5 --shard 3/8"
run() { timeout 600 $B $2 > $OUT/x.json 2> $OUT/x.err; python - $OUT/x.json "$1" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:52s} ms_per_step {j['ms_per_step']:8.4f}  Mrays/s {j['value']:8.1f}  hits {j['hits_sha256']}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run "product, learned tile order (default)" ""
run "product, traverse.tile_order=0" "--opts traverse.tile_order=0"
export HAGRID_AMD_LIB=$PWD/tools/proto/_work/libhagrid_amd.so
HG_PROTO_REFILL=2 run "prototype, refill 2 (no tile order)" ""
run "product again, traverse.tile_order=0" "--opts traverse.tile_order=0"
