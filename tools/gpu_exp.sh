#!/bin/bash
# One-off experiment (round 5, job 4): the general layout with links resolved inside the look-up (before the tests) against round 4's table layout on
# configuration 3's grid (A = this tree; B = the library before the general layout); traversal tests; the clustered bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 2400 python -m pytest tests/test_traverse_gpu.py -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log | cut -c1-300
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "clustered" > $OUT/pytest_clustered.log 2>&1; tail -3 $OUT/pytest_clustered.log | cut -c1-300
cp hagrid_amd/libhagrid_amd.so /tmp/libA.so
ab() {  # batch, env
  for round in 1 2; do for v in A B; do
    cp $( [ $v = A ] && echo /tmp/libA.so || echo ab/lib$v.so ) hagrid_amd/libhagrid_amd.so; touch hagrid_amd/libhagrid_amd.so
    echo -n "$v $2 | "; env $2 python tools/dev_option_sweep.py traverse.tail 1 --reps 1 --launches ${3:-100} --batch "$1" 2>&1 | tail -1 | cut -c1-200
  done; done
  cp /tmp/libA.so hagrid_amd/libhagrid_amd.so
}
ab "config3 4096^2" "X=1" 20
ab "primary 1024^2" "TD=0.15 SD=3.0"
ab "incoherent 4M binned" "TD=0.15 SD=3.0" 20
cp /tmp/libA.so hagrid_amd/libhagrid_amd.so
timeout 600 python tools/dev_nonuniform.py frames > $OUT/nonuniform_frames.txt 2>&1; cut -c1-500 $OUT/nonuniform_frames.txt
OPTS=traverse.image_slim=1 SD=5.0 timeout 600 python tools/dev_option_sweep.py traverse.tile_order 0 --batch "primary 4096^2" --reps 1 --launches 20 2>&1 | tail -1 | cut -c1-300
timeout 900 python bench.py --config clustered --steps 20 --warmup 3 > $OUT/bench_clustered.json 2> $OUT/bench_clustered.err; python - $OUT/bench_clustered.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); t = j["tile_order"] or {}
    print("clustered: ms_per_step", j["ms_per_step"], "Mrays/s", j["value"], "default order", t.get("ms_per_step_default_order"), "build_ms", j["build_ms"], "setup_ms", j["setup_traversal_ms"], "parity", j.get("parity"), "cpu", (j.get("cpu_baseline") or {}).get("value"))
    print(json.dumps(j["roofline"])[:600])
except Exception as e: print("FAILED", e, open(sys.argv[1][:-5] + ".err").read()[-600:])
PY
timeout 900 python bench.py --config clustered --rays aimed --steps 20 --warmup 3 --no-cpu-baseline --inflight 0 > $OUT/bench_clustered_aimed.json 2> $OUT/bench_clustered_aimed.err; python - $OUT/bench_clustered_aimed.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print("clustered aimed: ms_per_step", j["ms_per_step"], "Mrays/s", j["value"], "hit fraction", j["hit_fraction"])
except Exception as e: print("FAILED", e, open(sys.argv[1][:-5] + ".err").read()[-600:])
PY
