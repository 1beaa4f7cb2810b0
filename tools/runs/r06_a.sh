#!/bin/bash
# round 6, run A: fp8 tests after the FFN-subset switch; FFN-subset sensitivity study on both engineered rank sets
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_a
python -m pytest tests/test_fp8_gpu.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r06_a/pytest_fp8.txt
timeout 1500 python tools/fp8_ffn_subset_study.py both > gpurun_out/r06_a/fp8_ffn_subset_study.txt 2>&1
tail -4 gpurun_out/r06_a/pytest_fp8.txt; cat gpurun_out/r06_a/fp8_ffn_subset_study.txt | cut -c1-260
