#!/bin/bash
# DEV: per-kernel time of the construction passes (rocprofv3 kernel stats of a few builds) + build_ms
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-bp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$PWD
timeout 300 python -m pytest tests/test_build_gpu.py -x -q 2>&1 | tail -2
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from hagrid_amd import api, scene
mem = api.MemManager(keep=True); tris = scene.make_soup(1000000); d = mem.upload(tris)
g = api.build_all(mem, d, 1000000)
for comp in (False,):
    ts = []
    for _ in range(5):
        g.free(); ts.append(api.profile(lambda: api.build_all(mem, d, 1000000, grid=g), mem))
    print("build_ms", [round(t, 2) for t in ts])
import time
# stage breakdown
g.free()
for name, fn in (("build", lambda: api.build_grid(mem, d, 1000000, g, 0.12, 2.4)), ("merge", lambda: api.merge_grid(mem, g, 0.995)), ("flatten", lambda: api.flatten_grid(mem, g)), ("expand", lambda: api.expand_grid(mem, g, d, 3))):
    print(name, round(api.profile(fn, mem), 3), "ms")
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o trace -- python $ROOT/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --build-iter 5 > /dev/null 2> $ROOT/$OUT/prof.err)
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:28]:
    print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.3f}')
PY
find $OUT -name "*kernel_trace.csv" -delete
