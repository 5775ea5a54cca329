#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 68: the general layout's instantiations at eight resident wavefronts per SIMD (library B) against seven (A)
OUT=gpurun_out/r6geo; mkdir -p $OUT
cp hagrid_amd/libhagrid_amd.so /tmp/libA.so
for v in A B A B; do
  if [ $v = A ]; then cp /tmp/libA.so hagrid_amd/libhagrid_amd.so; else cp ab/libB.so hagrid_amd/libhagrid_amd.so; fi
  for sc in clustered stadium; do
    for b in "primary 1024^2" "primary 4096^2" "incoherent 4M binned"; do
      r=$(SCENE=$sc timeout 300 python tools/dev_option_sweep.py traverse.tile_order -1 --batch "$b" --reps 1 --launches 60 2>&1 | grep "ms_median" | cut -c50-110)
      echo "lib $v  $sc  $r"
    done
  done
done | tee $OUT/general_waves.txt
cp /tmp/libA.so hagrid_amd/libhagrid_amd.so
