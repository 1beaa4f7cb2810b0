#!/bin/bash
# round 6, run D: the 128 x 256 two-workgroups-per-CU GEMM against the automatic choice (tools/gemm2_ab.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_d
timeout 600 python tools/gemm2_ab.py 20 3 > gpurun_out/r06_d/gemm2_ab.txt 2>&1
cat gpurun_out/r06_d/gemm2_ab.txt | cut -c1-260
