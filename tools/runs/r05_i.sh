#!/bin/bash
# round 5, run I: full GPU suite + smoke on the sources after the K >= 1024 slicing gate; gemm_fp8_bench (epilogue decomposition)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_i
python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | cut -c1-400 | tail -80 > gpurun_out/r05_i/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_i/smoke.txt 2>&1
python tools/gemm_fp8_bench.py > gpurun_out/r05_i/gemm_fp8_bench.txt 2>&1
grep -n "FAILED\|passed\|failed" gpurun_out/r05_i/pytest.txt; tail -2 gpurun_out/r05_i/smoke.txt | cut -c1-200; cat gpurun_out/r05_i/gemm_fp8_bench.txt
