#!/bin/bash
# round 6, run C: the whole GPU suite (Phi-3 tower, fp8 presets, tightened per-element bounds), smoke, default bench with the new legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_c
python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | cut -c1-400 | tail -40 > gpurun_out/r06_c/pytest.txt
python -m pytest tests/test_phi3_gpu.py -m gpu -q -s -k "single_layer or full" 2>&1 | grep -v "^$" | cut -c1-600 | tail -40 > gpurun_out/r06_c/pytest_phi3_4b.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_c/smoke.txt 2>&1
python bench.py > gpurun_out/r06_c/bench_default.json 2> gpurun_out/r06_c/bench.err
grep -n "FAILED\|passed\|failed\|Error" gpurun_out/r06_c/pytest.txt | head -20; tail -12 gpurun_out/r06_c/pytest_phi3_4b.txt; tail -2 gpurun_out/r06_c/smoke.txt | cut -c1-200; tail -3 gpurun_out/r06_c/bench.err; python - <<'PY'
import json
p = json.loads(open('gpurun_out/r06_c/bench_default.json').read().strip().splitlines()[-1])
for k in ("value", "value_all_work", "ms_per_step", "frac_of_mfma_roofline", "frac_all_work", "executed_tflop_per_pair"):
    print(k, p.get(k))
print(p['roofline'])
for k, v in p['secondary'].items():
    print(k, {a: b for a, b in v.items() if a in ("value", "ms_per_step", "error", "leg_wall_s", "frac_of_mfma_roofline")})
c5 = p['secondary'].get('c5_4b', {})
for k in ("bf16", "fp8_ffn", "fp8_rank999"):
    if k in c5:
        print(k, c5[k].get("value"), c5[k].get("ms_per_step"), c5[k].get("frac_of_bf16_mfma_roofline"), c5[k].get("kernels"))
print({k: c5.get(k) for k in ("N", "parameters_G", "algorithmic_tflop_per_pair", "roofline_pairs_per_s_bf16")})
PY
