#!/bin/bash
# One-off experiment (round 5, job 13): the whole GPU suite after the prune; records of a uniform block in brick order (C = -DHG_BRICK) on the per-GPU share of
# configuration 5: time, hits, L2 misses per ray, with the caller's triangles and with triangles padded to 64 bytes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log | cut -c1-300
cp hagrid_amd/libhagrid_amd.so /tmp/libA.so
B="python bench.py --gpus 1 --steps 10 --warmup 3 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare --hits-hash --config 5 --shard 3/8"
export TMPDIR=/tmp; ROOT=$PWD
for v in A C; do
  cp $( [ $v = A ] && echo /tmp/libA.so || echo ab/lib$v.so ) hagrid_amd/libhagrid_amd.so; touch hagrid_amd/libhagrid_amd.so
  for pad in -1 1; do
    timeout 900 $B --opts traverse.tri_pad=$pad > $OUT/x.json 2> $OUT/x.err; python - $OUT/x.json "$v tri_pad=$pad" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:20s} ms_per_step {j['ms_per_step']:8.4f}  kernel_ms {j['roofline']['kernel_ms']:8.4f}  Mrays/s {j['value']:8.1f}  hits {j['hits_sha256'][:12]}  image {j['memory']['traversal_image']}")
except Exception as e: print(sys.argv[2], "FAILED", e, open(sys.argv[1][:-5] + ".err").read()[-400:])
PY
    (cd /tmp && PYTHONPATH=$ROOT timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_${v}_$pad -o pmc -- python $ROOT/bench.py --gpus 1 --steps 5 --warmup 2 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare --config 5 --shard 3/8 --opts traverse.tri_pad=$pad > $OUT/pmc_${v}_$pad.out 2> $OUT/pmc_${v}_$pad.err)
    python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_${v}_$pad/**/*counter_collection.csv", recursive=True)
if not f: print("no counters for $v pad $pad"); raise SystemExit
rows = [r for r in csv.DictReader(open(f[0])) if "traverse_kernel_tail" in r["Kernel_Name"]]
first = min(int(r["Dispatch_Id"]) for r in rows)
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if int(r["Dispatch_Id"]) == first: continue          # (the primary launch that produces the bounce rays)
    a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("$v tri_pad=$pad counters per ray:", {c: round(x / n / 8388608, 3) for c, (x, n) in sorted(acc.items())}, "launches", {c: n for c, (x, n) in acc.items()})
PY
    find $OUT/pmc_${v}_$pad -name "*.csv" -size +5M -delete
  done
done
cp /tmp/libA.so hagrid_amd/libhagrid_amd.so
