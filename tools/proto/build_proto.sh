#!/bin/bash
# Builds the PROTOTYPE library of tools/proto/ (a patched working copy of hagrid_amd/csrc under tools/proto/_work/, git-ignored) into
# tools/proto/_work/libhagrid_amd.so with the product's flags.  The product sources are not touched: bench / tools load the prototype through
# HAGRID_AMD_LIB (hagrid_amd/lib.py), which also keeps the test library away from it.
# usage: tools/proto/build_proto.sh            (first: cp -r hagrid_amd/csrc tools/proto/_work/csrc && patch -p3 -d tools/proto/_work/csrc < tools/proto/<name>.patch)
cd "$(dirname "$0")/../.." || exit 1
W=tools/proto/_work; mkdir -p $W/obj
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DHOST=__host__ -DDEVICE=__device__ -Iinclude -I$W/csrc -Wall -Wno-unused-function $PROTO_EXTRA"
pids=()
for s in $W/csrc/*.hip; do
  o=$W/obj/$(basename ${s%.hip}).o
  if [ ! -f $o ] || [ $s -nt $o ] || [ -n "$(find $W/csrc -maxdepth 1 -name '*.h' -newer $o 2>/dev/null | head -1)" ]; then
    /opt/rocm/bin/hipcc $F -c $s -o $o $( [ "$(basename $s)" = traverse.hip ] && echo "-Rpass-analysis=kernel-resource-usage" ) 2> $W/obj/$(basename ${s%.hip}).log & pids+=($!)
    if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]} || exit 1; pids=("${pids[@]:1}"); fi
  fi
done
for p in "${pids[@]}"; do wait $p || { echo "compile failed"; grep -h "error" $W/obj/*.log | head; exit 1; }; done
g++ -shared -fPIC -o $W/libhagrid_amd.so $W/obj/*.o -lpthread -ldl && echo "built $W/libhagrid_amd.so"
