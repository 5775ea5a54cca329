#!/bin/bash
# One-off experiment (round 5, job 15): the general layout's virtual top level (trav_image.hip image_general_vtop) -- traversal tests, then the clustered scene and
# configuration 3's grid forced into the general layout, with ("traverse.image_vtop" = 1, default) and without it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 1500 python -m pytest tests/test_traverse_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log | cut -c1-300
B="python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --inflight 0 --no-order-compare --hits-hash"
for round in 1 2; do
for vt in 1 0; do
  for what in "--config clustered" "--config clustered --rays aimed" "--config 3 --opts traverse.image_general=2,traverse.image_vtop=$vt"; do
    case "$what" in *opts*) W="$what";; *) W="$what --opts traverse.image_vtop=$vt";; esac
    timeout 900 $B $W > $OUT/x.json 2> $OUT/x.err; python - $OUT/x.json "vtop=$vt $what" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(f"{sys.argv[2][:70]:70s} ms_per_step {j['ms_per_step']:8.4f}  kernel_ms {j['roofline']['kernel_ms']:8.4f}  Mrays/s {j['value']:8.1f}  hits {j['hits_sha256'][:12]}  image {j['memory']['traversal_image']}")
except Exception as e: print(sys.argv[2], "FAILED", e, open(sys.argv[1][:-5] + ".err").read()[-600:])
PY
  done
done
done
