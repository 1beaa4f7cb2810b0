#!/bin/bash
# round 6, run F: tools/collect_profiles.sh r06_f - rocprofv3 kernel stats, the two PMC traffic passes, default / fp8 bench lines on the
# round's kernel sources (attention.hip with head_dim 96, gemm.hip with the two-per-CU kernel, rowops.hip with rope_heads)
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06_f > gpurun_out/collect_r06_f.log 2>&1
tail -5 gpurun_out/collect_r06_f.log | cut -c1-300; cat gpurun_out/prof_r06_f/pmc_summary.txt | head -30
