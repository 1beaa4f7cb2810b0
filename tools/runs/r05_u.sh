#!/bin/bash
# round 5, run U: the parity statements of the final sources as the tests print them (rank sets bf16 + mxfp8, full fixtures, stressed model, cache on / off)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_u
python -m pytest tests -m gpu -q -s -k "rank_agreement or full_c or stressed or prefix_cache or c3_shard or single_layer or oracle_c1_dims or tiny" 2>&1 | grep -v "^$\|^\.*$\|amdgpu.ids" | cut -c1-1200 > gpurun_out/r05_u/parity.txt
tail -3 gpurun_out/r05_u/parity.txt; wc -l gpurun_out/r05_u/parity.txt
python tools/parity_report.py full_c2 2>/dev/null | tail -15 >> gpurun_out/r05_u/parity.txt
