#!/bin/bash
# round 5, run D: the rest of the GPU suite after run C's stop at the mxfp8 prefix-cache bound; fused-vs-unfused table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_d
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r05_d/pytest.txt
python tools/fused_vs_unfused.py > gpurun_out/r05_d/fused_vs_unfused.txt 2>&1
tail -8 gpurun_out/r05_d/pytest.txt; cat gpurun_out/r05_d/fused_vs_unfused.txt
