#!/bin/bash
# DEV TOOL: builds a VARIANT of the product library for same-box A/B runs (tools/dev_ab.sh): ab/lib<NAME>.so from the sources under SRC (default: this
# tree's hagrid_amd/csrc + include; or an exported older tree: `git archive <commit> hagrid_amd/csrc include | tar -x -C /tmp/old`) with extra compiler
# flags.  ab/ is git-ignored and travels to the GPU box.   usage: tools/build_variant.sh NAME [SRC_ROOT] [-DHG_...]
cd "$(dirname "$0")/.." || exit 1
NAME=$1; SRC=${2:-.}; shift; shift
W=/tmp/hg_variant_$NAME; mkdir -p $W ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DHOST=__host__ -DDEVICE=__device__ -I$SRC/include -I$SRC/hagrid_amd/csrc -Wall -Wno-unused-function $*"
pids=()
for s in $SRC/hagrid_amd/csrc/*.hip; do
  o=$W/$(basename ${s%.hip}).o
  /opt/rocm/bin/hipcc $F -c $s -o $o 2> $W/$(basename ${s%.hip}).log & pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]} || { echo "compile failed"; grep -h "error" $W/*.log | head; exit 1; }; pids=("${pids[@]:1}"); fi
done
for p in "${pids[@]}"; do wait $p || { echo "compile failed"; grep -h "error" $W/*.log | head; exit 1; }; done
g++ -shared -fPIC -o ab/lib$NAME.so $W/*.o -lpthread -ldl && echo "built ab/lib$NAME.so"
