#!/bin/bash
# rocprofv3 kernel trace + stats of an arbitrary command; usage: tools/gpu_prof_cmd.sh TAG cmd...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- "$@" > $OUT/cmd.out 2> $OUT/cmd.err)
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:25]:
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.2f} total_ms {float(r["TotalDurationNs"])/1e6:9.3f}')
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
