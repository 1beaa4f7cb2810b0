#!/bin/bash
# round 6, run S: 2400 s of the extended fuzzer, a third seed, on the final kernel sources
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_s
timeout 2700 python tools/fuzz_kernels.py 2400 777001 > gpurun_out/r06_s/fuzz_2400s.txt 2>&1; tail -3 gpurun_out/r06_s/fuzz_2400s.txt | cut -c1-400
