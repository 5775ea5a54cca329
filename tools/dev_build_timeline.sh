#!/bin/bash
# DEV: kernel timeline of ONE construction (rocprofv3 --kernel-trace): per-launch durations in launch order + idle gaps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-bt}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/prof -o trace -- python $ROOT/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --build-iter 2 > $ROOT/$OUT/bench.json 2> $ROOT/$OUT/prof.err)
python - <<PY
import csv, glob, json
f = glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last complete construction: from the last bbox_partials to the first kernel after the last overlap_step
starts = [i for i, r in enumerate(rows) if "bbox_partials" in r["Kernel_Name"]]
b = starts[-1]
e = max(i for i, r in enumerate(rows) if "overlap_step" in r["Kernel_Name"] or "expand_" in r["Kernel_Name"]) + 1
seg = rows[b:e]
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"construction: {len(seg)} launches, span {(t1 - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
prev = t0
big = []
for r in seg:
    s, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("hagrid_impl::", "")[:44]
    big.append((name, (s - prev) / 1e3, (en - s) / 1e3)); prev = en
for name, gap, dur in big:
    if dur > 25 or gap > 15 or "expand_" in name: print(f"  gap {gap:7.1f} us  run {dur:7.1f} us  {name}")
print(open("$OUT/bench.json").read()[:0])
PY
find $OUT -name "*kernel_trace.csv" -delete
