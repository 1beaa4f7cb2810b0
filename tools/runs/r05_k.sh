#!/bin/bash
# round 5, run K: the whole GPU suite on the final sources; 600 s of the extended fuzzer (ABI 6 attention, K-sliced fp8 GEMMs); race screens
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_k
python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | cut -c1-400 | tail -60 > gpurun_out/r05_k/pytest.txt
timeout 900 python tools/fuzz_kernels.py 600 20261004 > gpurun_out/r05_k/fuzz_600s.txt 2>&1
grep -n "FAILED\|passed\|failed" gpurun_out/r05_k/pytest.txt; tail -12 gpurun_out/r05_k/fuzz_600s.txt | cut -c1-400
