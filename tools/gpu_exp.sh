#!/bin/bash
# One-off experiment (round 5, job 5): the in-place merge iterations on lists (construction tests + time); counters of the general layout (A) against round 4's
# table layout (B) on configuration 3's grid, 4096^2 primary rays.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 2400 python -m pytest tests/test_build_gpu.py tests/test_scan_gpu.py -x -q > $OUT/pytest_build.log 2>&1; tail -12 $OUT/pytest_build.log | cut -c1-400
timeout 300 python tools/dev_build_time.py 2>&1 | tail -1 | cut -c1-400
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "structure or build" > $OUT/pytest_full.log 2>&1; tail -3 $OUT/pytest_full.log | cut -c1-300
cp hagrid_amd/libhagrid_amd.so /tmp/libA.so
export TMPDIR=/tmp; ROOT=$PWD
for v in A B; do
  cp $( [ $v = A ] && echo /tmp/libA.so || echo ab/lib$v.so ) hagrid_amd/libhagrid_amd.so; touch hagrid_amd/libhagrid_amd.so
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAVES"; do
    i=$((i+1))
    (cd /tmp && PYTHONPATH=$ROOT timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/$OUT/$v$i -o pmc -- python $ROOT/tools/dev_option_sweep.py traverse.tail 1 --reps 1 --launches 10 --batch "config3 4096^2" > $ROOT/$OUT/$v$i.out 2> $ROOT/$OUT/$v$i.err)
    python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/$v$i/**/*counter_collection.csv", recursive=True)
if not f: print("no output for $v set $i"); raise SystemExit
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    if "traverse_kernel_tail" in r["Kernel_Name"] and int(r["Grid_Size"]) > 10000000:
        k = r["Counter_Name"]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for c, (x, n) in sorted(acc.items()): print(f"$v {c:32s} per ray {x/n/16777216:12.3f}  launches {n}")
PY
    find $OUT/$v$i -name "*.csv" -size +5M -delete
  done
done
cp /tmp/libA.so hagrid_amd/libhagrid_amd.so
