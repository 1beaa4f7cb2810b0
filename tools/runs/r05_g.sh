#!/bin/bash
# round 5, run G: stale-scratch probe (default switches; then each of the new switches off), the prefix / stressed tests again
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_g
for args in "" "prefix_cache=False" "trim_last_layer=False" "prefix_cache=False trim_last_layer=False" "norm_fusion=True"; do
  echo "== $args" >> gpurun_out/r05_g/probe.txt
  python tools/stale_scratch_probe.py $args 2>/dev/null >> gpurun_out/r05_g/probe.txt
done
python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "prefix_cache or stressed" 2>&1 | grep -v "^$" | cut -c1-1500 | grep -i "stressed tiny\|passed\|failed\|Error\|prefix cache on" > gpurun_out/r05_g/pytest.txt
cat gpurun_out/r05_g/probe.txt; cat gpurun_out/r05_g/pytest.txt
