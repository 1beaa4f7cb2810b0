#!/bin/bash
# round 6, run E: one / two / eight videos per forward, per kernel; K-slicing table for the under-filled GEMMs; tiny + kernel tests after
# the mask generalisation and the two-per-CU GEMM
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_e
python -m pytest tests/test_e2e_gpu.py tests/test_kernels_gpu.py tests/test_phi3_gpu.py -m gpu -q -x -k "tiny or two_per_cu or collated or random_batches" 2>&1 | tail -5 > gpurun_out/r06_e/pytest.txt
python tools/single_video_profile.py > gpurun_out/r06_e/single_video_profile.txt 2>&1
python tools/single_video_gemm_table.py > gpurun_out/r06_e/single_video_gemm_table.txt 2>&1
cat gpurun_out/r06_e/pytest.txt; cat gpurun_out/r06_e/single_video_profile.txt | grep -v amdgpu; cat gpurun_out/r06_e/single_video_gemm_table.txt | grep -v amdgpu | cut -c1-420
