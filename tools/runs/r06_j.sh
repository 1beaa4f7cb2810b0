#!/bin/bash
# round 6, run J: 900 s of the extended fuzzer on the final kernel sources (head_dim 96 attention, the two-per-CU GEMM, every older kernel)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_j
timeout 1100 python tools/fuzz_kernels.py 900 60606 > gpurun_out/r06_j/fuzz_900s.txt 2>&1; tail -3 gpurun_out/r06_j/fuzz_900s.txt | cut -c1-400
