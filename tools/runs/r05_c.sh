#!/bin/bash
# round 5, run C: full GPU suite (ABI 6, prefix cache, fp8 K-sliced tails), smoke, bench, per-Linear fp8 study
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_c
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r05_c/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_c/smoke.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_c/bench.json 2> gpurun_out/r05_c/bench.err
python bench.py --steps 20 --warmup 5 --fp8 > gpurun_out/r05_c/bench_fp8.json 2>> gpurun_out/r05_c/bench.err
timeout 900 python tools/fp8_per_linear_study.py > gpurun_out/r05_c/fp8_per_linear_study.txt 2>&1
tail -6 gpurun_out/r05_c/pytest.txt; tail -3 gpurun_out/r05_c/smoke.txt | cut -c1-300; head -c 400 gpurun_out/r05_c/bench.json; echo; head -c 400 gpurun_out/r05_c/bench_fp8.json; echo; cat gpurun_out/r05_c/fp8_per_linear_study.txt
