#!/bin/bash
# The timeline of one construction (kernel durations, gaps between dispatches) per scene: rocprofv3 --kernel-trace of tools/dev_build_time.py, condensed by
# tools/dev_build_timeline.py.   usage: tools/gpu_build_timeline.sh TAG [scene ...]   (scene: soup | clustered | stadium | ...)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" || exit 1
for sc in "${@:-soup}"; do
  S=$sc; [ $sc = soup ] && S=""
  (cd /tmp && SCENE=$S ITERS=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$sc -o trace -- python $OLDPWD/tools/dev_build_time.py > $OUT/build_time_$sc.txt 2> $OUT/build_time_$sc.err)
  tail -1 $OUT/build_time_$sc.txt | cut -c1-260
  T=$(find $OUT/prof_$sc -name "*kernel_trace.csv" | head -1)
  python tools/dev_build_timeline.py $T > $OUT/timeline_$sc.txt; python tools/dev_build_timeline.py $T --all > $OUT/timeline_all_$sc.txt; cat $OUT/timeline_$sc.txt
  S2=$(find $OUT/prof_$sc -name "*kernel_stats.csv" | head -1); [ -n "$S2" ] && cp $S2 $OUT/kernel_stats_$sc.csv
  rm -rf $OUT/prof_$sc
done
