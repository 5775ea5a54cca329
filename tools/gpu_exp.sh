#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 12: construction parity + timelines after the incremental walk in count_top_refs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_build_gpu.py -m gpu -q -x 2>&1 | tail -2 | cut -c1-300
bash tools/gpu_build_timeline.sh r6l soup clustered 2>&1 | grep "build_ms_mean\|one construction" | cut -c1-160
python - <<'PY'
import csv
for sc in ("soup","clustered"):
    for r in csv.DictReader(open(f"gpurun_out/r6l/kernel_stats_{sc}.csv")):
        if "count_top_refs" in r["Name"] or "emit_top_refs" in r["Name"]:
            print(sc, r["Name"][23:40], r["Calls"], "avg", round(float(r["AverageNs"])/1e3,1), "min", round(float(r["MinNs"])/1e3,1))
PY
