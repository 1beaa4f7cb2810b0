#!/bin/bash
# round 5, run E: prefix cache through a prefix-only pass (cold = warm); full GPU suite, stress statistics, fused-vs-unfused again
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_e
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r05_e/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_e/smoke.txt 2>&1
timeout 900 python tools/stress_stats.py > gpurun_out/r05_e/stress_stats.txt 2>&1
python tools/fused_vs_unfused.py 2>/dev/null > gpurun_out/r05_e/fused_vs_unfused.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_e/bench.json 2> gpurun_out/r05_e/bench.err
tail -4 gpurun_out/r05_e/pytest.txt | cut -c1-300; grep -i "stressed tiny" gpurun_out/r05_e/pytest.txt | cut -c1-900; cat gpurun_out/r05_e/stress_stats.txt; cat gpurun_out/r05_e/fused_vs_unfused.txt; head -c 300 gpurun_out/r05_e/bench.json
