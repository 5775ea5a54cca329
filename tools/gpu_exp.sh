#!/bin/bash
# One-off experiment (round 5, job 12): after the prune (slim layouts only) -- the whole GPU suite, smoke, the headline line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 3000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench2.json 2> $OUT/bench2.err; python - $OUT/bench2.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); t = j["tile_order"]
    print("config 2: value", j["value"], "ms_per_step", j["ms_per_step"], "default order", t["ms_per_step_default_order"], "build_ms", j["build_ms"], "setup_ms", j["setup_traversal_ms"], "parity", j["parity"], "cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
    print({k: (v["ms_per_frame"], v["ms_per_frame_default_order"]) for k, v in t["moving_camera"].items() if isinstance(v, dict)})
    print(j["config"]["traversal_image"])
except Exception as e: print("FAILED", e, open(sys.argv[1][:-5] + ".err").read()[-600:])
PY
