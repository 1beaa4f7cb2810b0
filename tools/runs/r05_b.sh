#!/bin/bash
# round 5, run B: ABI 6 (suffix queries, shared prefix), last-layer query trimming, prefix cache: targeted tests, then smoke + bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_b
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "suffix or nonfinite or gelu or rope or test_attention" 2>&1 | tail -15 > gpurun_out/r05_b/pytest_kernels.txt
python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -s -k "prefix_cache or trimming or tiny or full_c or sticky or batch" 2>&1 | tail -40 > gpurun_out/r05_b/pytest_e2e.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_b/smoke.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_b/bench.json 2> gpurun_out/r05_b/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prefix-cache --no-secondary > gpurun_out/r05_b/bench_nocache.json 2>> gpurun_out/r05_b/bench.err
tail -5 gpurun_out/r05_b/pytest_kernels.txt; tail -12 gpurun_out/r05_b/pytest_e2e.txt; tail -4 gpurun_out/r05_b/smoke.txt; head -c 600 gpurun_out/r05_b/bench.json; echo; head -c 300 gpurun_out/r05_b/bench_nocache.json
