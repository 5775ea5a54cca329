#!/bin/bash
# One-off measurements (end of round 4, no source changes: options only).  (1) construction of configuration 3's grid and of the 8M-triangle grid with and without the
# expansion's resolved voxel map; (2) the per-GPU share of configuration 5 with the triangles padded to 64 bytes (kernel time apart from the copy: rocprofv3 stats);
# (3) per-launch times of the construction kernels of one build (which level costs classify_refs what).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp3}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
B="python $PWD/bench.py --gpus 1 --steps 5 --warmup 2 --build-iter 10 --no-cpu-baseline --inflight 0 --no-order-compare"
pick() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "| build_ms", j["build_ms"], "min", j.get("build_ms_min"), "| ms_per_step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"], "value", j["value"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for rep in ${REPS:-1 2}; do
  for vm in 1 0; do
    timeout 600 $B --config 3 --opts expand.voxel_map=$vm > $OUT/c3_vm${vm}_$rep.json 2> $OUT/c3_vm${vm}_$rep.err; pick $OUT/c3_vm${vm}_$rep.json "config 3, voxel_map $vm, rep $rep"
  done
done
for vm in ${VM5:-1 0}; do
  timeout 900 $B --config 5 --shard 3/8 --build-iter 5 --opts expand.voxel_map=$vm > $OUT/c5_vm$vm.json 2> $OUT/c5_vm$vm.err; pick $OUT/c5_vm$vm.json "config 5 share, voxel_map $vm"
done
export TMPDIR=/tmp; ROOT=$PWD
for pad in 0 1; do
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/pad$pad -o trace -- $B --build-iter 1 --config 5 --shard 3/8 --opts traverse.tri_pad=$pad > $ROOT/$OUT/c5_pad$pad.json 2> $ROOT/$OUT/c5_pad$pad.err)
  pick $OUT/c5_pad$pad.json "config 5 share, tri_pad $pad"
  python - $OUT/pad$pad <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:40] if f else []:
    if "traverse_kernel_tail" in r["Name"] or "pad_triangles" in r["Name"]:
        print("   ", r["Name"][:80], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2))
PY
done
(cd /tmp && ITERS=2 PYTHONPATH=$ROOT timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/trace -o trace -- python $ROOT/tools/dev_build_time.py > $ROOT/$OUT/trace.out 2> $ROOT/$OUT/trace.err)
python - $OUT/trace <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"])) if f else []
# the last build of the run: from the last bbox_partials on
last = max((i for i, r in enumerate(rows) if "bbox_partials" in r["Kernel_Name"]), default=0)
per = collections.defaultdict(list)
for r in rows[last:]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    per[n].append(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1))
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print(f"{n:40s} total {sum(v):8.1f} us  calls {v}")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; du -sh $OUT
