#!/bin/bash
# round 6, run L: what was committed after run I, on the GPU: the long-context head_dim 96 test and bench.py --backbone 4b (bf16, both fp8 presets)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_l
python -m pytest tests/test_phi3_gpu.py -m gpu -q 2>&1 | grep -v "^$" | cut -c1-400 | tail -15 > gpurun_out/r06_l/pytest_phi3.txt
python bench.py --backbone 4b > gpurun_out/r06_l/bench_4b_bf16.json 2> gpurun_out/r06_l/bench_4b_bf16.err
python bench.py --backbone 4b --fp8 --no-secondary > gpurun_out/r06_l/bench_4b_fp8.json 2> gpurun_out/r06_l/bench_4b_fp8.err
python bench.py --backbone 4b --fp8 --fp8-preset mxfp8-rank999 --no-secondary > gpurun_out/r06_l/bench_4b_fp8_rank999.json 2> gpurun_out/r06_l/bench_4b_fp8_rank999.err
tail -3 gpurun_out/r06_l/pytest_phi3.txt; tail -c 600 gpurun_out/r06_l/*.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_l/bench_4b_*.json')):
    try:
        p = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'NO JSON', e); continue
    print(f, {k: p.get(k) for k in ("value", "ms_per_step", "frac_of_mfma_roofline", "dtype")}, p.get('roofline'))
    for k, v in sorted(p.get('kernels', {}).items(), key=lambda kv: -kv[1].get('ms_per_step', 0)):
        print('   ', k, v)
PY
