#!/bin/bash
# Construction tests + the construction timing of the 1M-triangle scene on the GPU box (tools/dev_build_time.py).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2l
timeout 1200 python -m pytest tests/test_build_gpu.py tests/test_scan_gpu.py -x -q > gpurun_out/r2l/pytest.log 2>&1; tail -3 gpurun_out/r2l/pytest.log
ITERS=${ITERS:-30} timeout 600 python tools/dev_build_time.py 2>/dev/null | tee gpurun_out/r2l/build_time2.log
