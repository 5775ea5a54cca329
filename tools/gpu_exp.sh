#!/bin/bash
# One-off experiment (round 4): the headline launch with this library, with the one before the last change (ab/libHEADgate.so) and with round 3's (ab/libR3.so: commit aef0877, ABI number patched), same box, alternating.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
B="python bench.py --gpus 1 --steps 20 --warmup 3 --build-iter 3 --no-cpu-baseline --inflight 0"
for rep in 1 2; do
  for v in HEAD HEADgate R3; do
    if [ $v = HEAD ]; then unset HAGRID_AMD_LIB; else export HAGRID_AMD_LIB=$PWD/ab/lib$v.so; fi
    timeout 600 $B > $OUT/${v}_$rep.json 2> $OUT/${v}_$rep.err
    python - $OUT/${v}_$rep.json "$v rep $rep" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); t = j.get("tile_order") or {}
    print(sys.argv[2], "| ms_per_step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"], "default_order", t.get("ms_per_step_default_order"), "build_ms", j["build_ms"])
except Exception as e: print(sys.argv[2], "FAILED", e, open(sys.argv[1][:-5] + ".err").read()[-600:])
PY
  done
done
unset HAGRID_AMD_LIB
timeout 900 python -m pytest tests/test_traverse_gpu.py tests/test_fullsize_gpu.py -x -q -k "tile or loop_over_one_buffer or row_length" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python tools/dev_moving_camera.py --speeds 0,1 --frames 40 > $OUT/moving.txt 2>&1; cut -c1-260 $OUT/moving.txt
