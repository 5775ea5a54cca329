#!/bin/bash
# One-off experiment (round 5, job 8): the merge of one construction, launch by launch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
export TMPDIR=/tmp; ROOT=$PWD
(cd /tmp && PYTHONPATH=$ROOT timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o trace -- python $ROOT/tools/dev_build_time.py > $OUT/cmd.out 2> $OUT/cmd.err)
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
# the last construction: from the last bbox_partials on
last = max(i for i, r in enumerate(rows) if "bbox_partials" in r["Kernel_Name"])
t0 = None
for r in rows[last:]:
    n = r["Kernel_Name"]
    if "expand" in n or "overlap" in n or "fill_voxel" in n: break
    if not any(k in n for k in ("ip_", "merge", "remap", "cell_flags", "scan", "widen")): continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9s}  {n.replace('(anonymous namespace)::', '')[:70]}")
PY
find $OUT -name "*kernel_trace.csv" -delete
