#!/bin/bash
# round 5, run F: the five failures of run E with full output; attention tests + stress statistics after the offset headroom
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_f
python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "trimming or prefix_cache or stressed or c3_shard" 2>&1 | grep -v "^$" | cut -c1-600 | tail -150 > gpurun_out/r05_f/pytest_failing.txt
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention or sweep" 2>&1 | tail -8 > gpurun_out/r05_f/pytest_attention.txt
timeout 900 python tools/stress_stats.py > gpurun_out/r05_f/stress_stats.txt 2>&1
grep -n "Error\|assert\|FAILED\|passed\|failed" gpurun_out/r05_f/pytest_failing.txt | head -60; tail -4 gpurun_out/r05_f/pytest_attention.txt; tail -12 gpurun_out/r05_f/stress_stats.txt
