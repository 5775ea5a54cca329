#!/bin/bash
# the whole GPU suite once more at the committed kernel sources (a test was added after the evidence run)
python -c "from hagrid_amd.build import source_hash; print('kernel sources', source_hash())" > gpurun_out/final_pytest_gpu_tail.txt
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 >> gpurun_out/final_pytest_gpu_tail.txt
cat gpurun_out/final_pytest_gpu_tail.txt
