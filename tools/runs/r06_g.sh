#!/bin/bash
# round 6, run G: the whole GPU suite + smoke + default bench on the final tree (the traffic figure now comes from profiles/r06_pmc_traffic.json)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_g
python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | cut -c1-400 | tail -30 > gpurun_out/r06_g/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_g/smoke.txt 2>&1
python bench.py > gpurun_out/r06_g/bench_default.json 2> gpurun_out/r06_g/bench.err
grep -n "FAILED\|passed\|failed\|Error" gpurun_out/r06_g/pytest.txt | head; tail -2 gpurun_out/r06_g/smoke.txt | cut -c1-160; python - <<'PY'
import json
p = json.loads(open('gpurun_out/r06_g/bench_default.json').read().strip().splitlines()[-1])
print({k: p.get(k) for k in ("value", "value_all_work", "ms_per_step", "frac_of_mfma_roofline", "frac_all_work")})
print(p['roofline'])
print({k: v.get('value') for k, v in p['secondary'].items()})
PY
