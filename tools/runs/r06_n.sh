#!/bin/bash
# round 6, run N: the whole GPU suite + smoke (now with the Phi-3 tower) + default bench on the final tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_n
python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | cut -c1-400 | tail -30 > gpurun_out/r06_n/pytest.txt
python -m pytest tests/test_phi3_gpu.py -m gpu -q -s -k "full or single_layer" 2>&1 | grep -v "^$" | cut -c1-600 | tail -60 > gpurun_out/r06_n/pytest_phi3_4b.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_n/smoke.txt 2>&1
python bench.py > gpurun_out/r06_n/bench_default.json 2> gpurun_out/r06_n/bench.err
grep -n "FAILED\|passed\|failed\|Error" gpurun_out/r06_n/pytest.txt | head; grep -n "passed\|failed" gpurun_out/r06_n/pytest_phi3_4b.txt; tail -3 gpurun_out/r06_n/smoke.txt | cut -c1-120; python - <<'PY'
import json
p = json.loads(open('gpurun_out/r06_n/bench_default.json').read().strip().splitlines()[-1])
print({k: p.get(k) for k in ("value", "value_all_work", "ms_per_step", "frac_of_mfma_roofline", "frac_all_work")})
print(p['roofline'])
print({k: v.get('value') for k, v in p['secondary'].items()})
PY
