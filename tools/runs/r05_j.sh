#!/bin/bash
# round 5, run J: the round's profile collection (rocprofv3 kernel stats, PMC traffic passes, default / fp8 bench lines) + the two fp8 tests touched last
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r05_j > gpurun_out/collect_r05_j.log 2>&1
python -m pytest tests/test_fp8_gpu.py -m gpu -q -k "tiny or sliced" 2>&1 | tail -3
tail -5 gpurun_out/collect_r05_j.log | cut -c1-400
