#!/bin/bash
# the kernel fuzzer on the committed library (every instantiation against the construction-format kernel, 20 random scenes)
timeout 900 python tools/dev_fuzz_kernels.py 2>&1 | tail -3 | cut -c1-300
