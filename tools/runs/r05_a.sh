#!/bin/bash
# round 5, run A: full GPU suite + smoke + default bench on the first sources of the round
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_a
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_a/pytest.txt
python __graft_entry__.py smoke > gpurun_out/r05_a/smoke.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_a/bench.json 2> gpurun_out/r05_a/bench.err
python -m pytest tests/test_fp8_gpu.py -m gpu -q -s -k "engineered" 2>&1 | grep -i "mxfp8 FFN vs" > gpurun_out/r05_a/fp8_rank.txt
tail -3 gpurun_out/r05_a/pytest.txt; cat gpurun_out/r05_a/smoke.txt | tail -4; head -c 1500 gpurun_out/r05_a/bench.json
