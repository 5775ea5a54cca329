#!/bin/bash
# One-off experiment script of round 6 (rewritten per job; outputs quoted in profiles/NOTES.md).  Job 4: the share trial -- frame loops at the viewer's speed and back-to-back
# launches in the default order on four scene families; tests that touch the policy.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6d; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 400 python -m pytest tests/test_traverse_gpu.py -m gpu -q -x -k "tile_order or head_share or lifetime or tile_packets" 2>&1 | tail -3 | cut -c1-250
for sc in clustered soup gradient shell stadium; do
  timeout 300 python tools/dev_frame_policies.py --scene $sc --speed 1.0 2>&1 | grep -v amdgpu | tee $OUT/policies_$sc.txt | cut -c1-600
done
for sc in clustered "" shell stadium; do
  echo "== back to back, default order, scene ${sc:-soup}"
  SCENE=$sc OPTS=traverse.tile_order=0 timeout 200 python tools/dev_option_sweep.py traverse.share_trial 0,1 --batch "primary 1024^2" --reps 2 --launches 100 2>&1 | grep ms_median | cut -c1-200 | tee -a $OUT/share_trial_${sc:-soup}.txt
done
