#!/bin/bash
# PROTOTYPE measurement (tools/proto/README.md): the per-GPU shares of configurations 4 and 5 with the prototype library (tools/proto/_work/libhagrid_amd.so), lanes refilled
# from a pool of K tiles per wavefront (HG_PROTO_REFILL = K) against the same library without (the product's path); the hit buffers must be the same bytes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-proto}; mkdir -p $OUT
export HAGRID_AMD_LIB=$PWD/tools/proto/_work/${PROTO_LIB:-libhagrid_amd.so}
[ -f $HAGRID_AMD_LIB ] || { echo "no prototype library"; exit 1; }
for C in ${CONFIGS:-4 5}; do
  for K in ${KS:-0 8 4 16}; do
    if [ $K = 0 ]; then unset HG_PROTO_REFILL; else export HG_PROTO_REFILL=$K; fi
    timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare --hits-hash --config $C --shard 3/8 > $OUT/c${C}_k$K.json 2> $OUT/c${C}_k$K.err
    python - $OUT/c${C}_k$K.json "config $C share, refill $K" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:28s} ms_per_step {j['ms_per_step']:8.4f}  kernel_ms {j['roofline']['kernel_ms']:8.4f}  Mrays/s {j['value']:8.1f}  hits {j['hits_sha256']}  hit_fraction {j['hit_fraction']}")
except Exception as e: print(sys.argv[2], "FAILED", e, open(sys.argv[1][:-5] + ".err").read()[-400:])
PY
  done
done
