#!/bin/bash
# DEV TOOL: same-box A/B of two builds of libhagrid_amd.so.  A = the library in the tree, B = ab/libB.so (built from an edited tree, see
# tools/README.md).  usage (inside one gpurun call): tools/dev_ab.sh "primary 1024^2" ["primary 640x480" ...]
set -u
cd "$(dirname "$0")/.."
cp hagrid_amd/libhagrid_amd.so /tmp/libA.so
for batch in "$@"; do
  for round in 1 2; do
    for v in A B; do
      if [ $v = A ]; then cp /tmp/libA.so hagrid_amd/libhagrid_amd.so; else cp ab/libB.so hagrid_amd/libhagrid_amd.so; fi
      touch hagrid_amd/libhagrid_amd.so
      echo -n "$v "; python tools/dev_option_sweep.py traverse.tail 1 --reps 1 --batch "$batch" 2>&1 | tail -1
    done
  done
done
cp /tmp/libA.so hagrid_amd/libhagrid_amd.so
