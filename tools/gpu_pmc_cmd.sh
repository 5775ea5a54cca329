#!/bin/bash
# DEV: PMC counters of the traversal kernels for an arbitrary command.
# usage: tools/gpu_pmc_cmd.sh TAG "python tools/dev_ab_traverse.py 2,5 3" "COUNTERS A" "COUNTERS B" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; CMD=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
i=0
for set in "$@"; do
  i=$((i+1))
  (cd /tmp && PYTHONPATH=$ROOT timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/$OUT/p$i -o pmc -- $CMD > $ROOT/$OUT/p$i.out 2> $ROOT/$OUT/p$i.err)
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/p$i/**/*counter_collection.csv", recursive=True)
if not f: print("no output for set $i: $set"); raise SystemExit
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    if "traverse_kernel" in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("traverse_kernel")[1][:24], r["Grid_Size"], r["Counter_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, gs, c), (v, n) in sorted(acc.items()):
    print(f"{kn:26s} grid {gs:>10s} {c:28s} avg/launch {v/n:16.1f}  launches {n}")
PY
  find $OUT/p$i -name "*.csv" -size +5M -delete
done
