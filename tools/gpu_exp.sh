#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 32: SPLIT with two-level counting.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.build()" || exit 1
for k in 20 14 10 6; do
  echo "== split_after $k (soup, clustered, stadium)"
  for sc in "" clustered stadium; do
  SCENE=$sc OPTS=traverse.tile_order=0,traverse.share_trial=0,traverse.split_after=$k timeout 120 python tools/dev_option_sweep.py traverse.split 1 --batch "primary 1024^2" --reps 1 --launches 60 2>&1 | grep "ms_median\|rror" | cut -c1-170
  done
done
