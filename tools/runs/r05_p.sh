#!/bin/bash
# round 5, run P: 1500 s of the extended fuzzer on the final sources (another seed) + the race screens of the kernel suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_p
timeout 1800 python tools/fuzz_kernels.py 1500 777 > gpurun_out/r05_p/fuzz_1500s_seed777.txt 2>&1
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "race or stability or repeat" 2>&1 | tail -3 > gpurun_out/r05_p/race_screens.txt
tail -5 gpurun_out/r05_p/fuzz_1500s_seed777.txt | cut -c1-400; cat gpurun_out/r05_p/race_screens.txt
