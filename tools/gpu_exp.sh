#!/bin/bash
# One-off experiment (round 5, job 23): scalar read-backs through a publishing wavefront ("ctx.fast_readback" = 1 / 0), short lists sorted in registers by the
# concatenation, the top-level cells written by the segment scan: the whole GPU suite, construction times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-exp}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 1500 python -m pytest tests/test_build_gpu.py tests/test_scan_gpu.py tests/test_concurrency_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log | cut -c1-300
for r in 1 2; do for f in 1 0; do echo "fast_readback=$f"; OPTS=ctx.fast_readback=$f ITERS=10 timeout 300 python tools/dev_build_pool.py 2>&1 | tail -1 | cut -c1-400; done; done
for f in 1 0; do echo "fast_readback=$f"; OPTS=ctx.fast_readback=$f timeout 300 python tools/dev_build_time.py 2>&1 | tail -1 | cut -c1-300; done
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_all.log 2>&1; tail -5 $OUT/pytest_all.log | cut -c1-300
