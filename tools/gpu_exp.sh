#!/bin/bash
# One-off experiment (round 5, job 30): the head share measures itself and stays once taken up: five scene families, traversal tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.build()" || exit 1
SCENE=clustered HAGRID_TRACE_HEAD=1 timeout 300 python tools/dev_option_sweep.py traverse.quad_head 20 --batch "primary 1024x1024" --reps 1 --launches 100 2>&1 | grep "head\]\|ms_median" | head -12 | cut -c1-200
run() { timeout 400 python tools/dev_option_sweep.py traverse.quad_head $1 --batch "$2" --reps 2 --launches 100 2>&1 | python -c "
import sys, json, collections
acc = collections.defaultdict(list); crc = set(); b = ''
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    if 'grid' in j: continue
    acc[j['traverse.quad_head']].append(j['ms_median']); crc.add(j['hits_crc']); b = j['batch']
print('$3', b, {k: [round(x, 4) for x in v] for k, v in acc.items()}, 'crc', len(crc))
"; }
for sc in clustered gradient shell; do
  for b in "primary 1024x1024" "primary 1536x1536" "primary 1920x1080"; do SCENE=$sc run 0,20 "$b" $sc; done
done
for b in "primary 1024x1024" "primary 2048x2048"; do run 0,20 "$b" uniform; done
timeout 1200 python -m pytest tests/test_traverse_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "not config5 and not config4 and not config3" 2>&1 | tail -3
