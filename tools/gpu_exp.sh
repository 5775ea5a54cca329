#!/bin/bash
# One-off experiment (round 5, job 3): the general layout against round 4's table layout on configuration 3's grid (A = this tree, 7 wavefronts per SIMD;
# B = the library before the general layout; C = this tree at 8 wavefronts per SIMD with spills); a launch without history: lanes refilled from a pool in the
# default tile order; the moving camera with the new give-up rule.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
cp hagrid_amd/libhagrid_amd.so /tmp/libA.so
ab() {  # batch, env
  for round in 1 2; do for v in A B C; do
    cp $( [ $v = A ] && echo /tmp/libA.so || echo ab/lib$v.so ) hagrid_amd/libhagrid_amd.so; touch hagrid_amd/libhagrid_amd.so
    echo -n "$v $2 | "; env $2 python tools/dev_option_sweep.py traverse.tail 1 --reps 1 --launches ${3:-100} --batch "$1" 2>&1 | tail -1 | cut -c1-200
  done; done
  cp /tmp/libA.so hagrid_amd/libhagrid_amd.so
}
ab "config3 4096^2" "X=1" 20
ab "primary 1024^2" "TD=0.15 SD=3.0"
ab "primary 2048^2" "TD=0.15 SD=3.0" 50
ab "incoherent 4M binned" "TD=0.15 SD=3.0" 20
cp /tmp/libA.so hagrid_amd/libhagrid_amd.so
for b in "primary 1024^2" "primary 1280x720" "primary 1920x1080" "primary 640x480"; do
  OPTS=traverse.tile_order=0 timeout 600 python tools/dev_option_sweep.py traverse.refill 0,2,3 --batch "$b" --reps 2 2>&1 | grep -v '"grid"' | cut -c1-200
done
timeout 600 python tools/dev_moving_camera.py --speeds 0,0.25,1 --frames 40 > $OUT/moving.txt 2>&1; cut -c1-300 $OUT/moving.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench2.json 2> $OUT/bench2.err; python - $OUT/bench2.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); t = j["tile_order"]
print("config 2: ms_per_step", j["ms_per_step"], "default order", t["ms_per_step_default_order"], "build_ms", j["build_ms"], "setup_ms", j["setup_traversal_ms"])
print(json.dumps(t.get("moving_camera"))[:900])
PY
