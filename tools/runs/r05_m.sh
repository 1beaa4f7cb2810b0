#!/bin/bash
# round 5, run M: e2e suite after the prefix cache's thrash guard
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_m
python -m pytest tests/test_e2e_gpu.py -m gpu -q 2>&1 | grep -v "^$" | cut -c1-300 | tail -40 > gpurun_out/r05_m/pytest.txt
grep -n "FAILED\|passed\|failed\|Error" gpurun_out/r05_m/pytest.txt | head
