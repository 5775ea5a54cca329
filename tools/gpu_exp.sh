#!/bin/bash
mkdir -p gpurun_out/exp
timeout 900 python tools/dev_fuzz_kernels.py 2>&1 | tail -20 > gpurun_out/exp/fuzz_kernels.txt
cat gpurun_out/exp/fuzz_kernels.txt | cut -c1-300
