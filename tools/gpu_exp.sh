#!/bin/bash
# One-off experiment script of round 6 (rewritten per job).  Job 46: host time of one hagrid_traverse_grid call on an idle stream
python tools/dev_host_time.py 2>&1 | grep -v amdgpu
