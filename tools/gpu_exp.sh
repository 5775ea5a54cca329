#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Last job: the tree as committed -- smoke, the default bench line (must accept the committed counter files), the policy tests
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/final_bench.json'))
print(d["value"], d["unit"], d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"].get("traffic_source", "")[:80], "kernel_sources", d["roofline"].get("kernel_sources"))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], "parity", d.get("parity", {}).get("hits_identical_to_oracle", d.get("parity")))
PY
timeout 900 python -m pytest tests/test_traverse_gpu.py -m gpu -q -x -k "share_trial or policy_state or tile_order" 2>&1 | tail -2
