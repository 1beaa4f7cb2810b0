#!/bin/bash
# round 5, run S: the whole GPU suite + smoke + default bench on the final tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s
python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | cut -c1-300 | tail -30 > gpurun_out/r05_s/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_s/smoke.txt 2>&1
python bench.py > gpurun_out/r05_s/bench_default.json 2> gpurun_out/r05_s/bench.err
grep -n "FAILED\|passed\|failed" gpurun_out/r05_s/pytest.txt; tail -2 gpurun_out/r05_s/smoke.txt | cut -c1-160; head -c 400 gpurun_out/r05_s/bench_default.json; echo; python - <<'PY'
import json
p = json.loads(open('gpurun_out/r05_s/bench_default.json').read().strip().splitlines()[-1])
print(p['roofline'])
print({k: v.get('value') for k, v in p['secondary'].items()})
PY
