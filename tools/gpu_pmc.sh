#!/bin/bash
# DEV: PMC diagnosis of the traversal kernel. usage: tools/gpu_pmc.sh TAG "COUNTERS A" "COUNTERS B" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
[ -f $OUT/counters.txt ] || (cd /tmp && rocprofv3 -L > $ROOT/$OUT/counters.txt 2>&1)
i=0
for set in "$@"; do
  i=$((i+1))
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/$OUT/p$i -o pmc -- python $ROOT/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --build-iter 1 --no-order-compare ${BENCH_ARGS} > /dev/null 2> $ROOT/$OUT/p$i.err)
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/p$i/**/*counter_collection.csv", recursive=True)
if not f: print("no output for set $i: $set"); raise SystemExit
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    if "traverse_kernel" in r["Kernel_Name"] and "Lb1" not in r["Kernel_Name"].split("traverse_kernel")[1][:12].replace("ILb0ELb1","STATS") or True:
        if "traverse_kernel" in r["Kernel_Name"]:
            k = (r["Kernel_Name"][:60], r["Counter_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, c), (v, n) in sorted(acc.items()):
    print(f"{kn:60s} {c:28s} avg/launch {v/n:16.1f}  launches {n}")
PY
  find $OUT/p$i -name "*.csv" -size +5M -delete
done
