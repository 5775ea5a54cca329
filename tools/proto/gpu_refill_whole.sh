#!/bin/bash
# PROTOTYPE measurement: whole batches (configuration 3: 16M primary rays; configuration 5: 64M bounce rays) with and without the refill of tools/proto/refill.patch
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-proto6}; mkdir -p $OUT
export HAGRID_AMD_LIB=$PWD/tools/proto/_work/libhagrid_amd.so
for C in ${CONFIGS:-3 5}; do
  for K in ${KS:-0 2}; do
    if [ $K = 0 ]; then unset HG_PROTO_REFILL; else export HG_PROTO_REFILL=$K; fi
    timeout 500 python bench.py --gpus 1 --steps 10 --warmup 3 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare --hits-hash --config $C > $OUT/c${C}_k$K.json 2> $OUT/c${C}_k$K.err
    python - $OUT/c${C}_k$K.json "config $C whole batch, refill $K" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:32s} ms_per_step {j['ms_per_step']:8.4f}  Mrays/s {j['value']:8.1f}  hits {j['hits_sha256']}")
except Exception as e: print(sys.argv[2], "FAILED", e, open(sys.argv[1][:-5] + ".err").read()[-300:])
PY
  done
done
