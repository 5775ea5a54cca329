#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 7: stadium test again; table-layout instantiations at seven resident wavefronts without spills (B) against
# eight with their spills around the loops (A), same box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6g; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "stadium" 2>&1 | tail -5 | cut -c1-300
echo "== config 3's grid, 1024^2 (learned order: cost bookkeeping)"; TD=0.15 SD=3.0 tools/dev_ab.sh "primary 1024^2" 2>&1 | grep -v amdgpu | cut -c1-200 | tee $OUT/ab_table_cost.txt
echo "== config 3's grid, 4096^2 (no costs: not affected -- control)"; tools/dev_ab.sh "config3 4096^2" 2>&1 | grep -v amdgpu | cut -c1-200 | tee $OUT/ab_table_plain.txt
echo "== soup at snd-density 5, 4096^2 (wide records)"; TD=0.12 SD=5.0 tools/dev_ab.sh "primary 4096^2" 2>&1 | grep -v amdgpu | cut -c1-200 | tee $OUT/ab_table_wide.txt
echo "== soup at snd-density 5, 1024^2 (wide records + costs)"; TD=0.12 SD=5.0 tools/dev_ab.sh "primary 1024^2" 2>&1 | grep -v amdgpu | cut -c1-200 | tee $OUT/ab_table_wide_cost.txt
