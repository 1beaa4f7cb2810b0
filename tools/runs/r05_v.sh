#!/bin/bash
# round 5, run V: fp8 GELU epilogue in the split-table form: fp8 tests, gemm_fp8_bench, bench --fp8
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_v
python -m pytest tests/test_fp8_gpu.py -m gpu -q -x 2>&1 | tail -12 | cut -c1-300 > gpurun_out/r05_v/pytest.txt
python tools/gemm_fp8_bench.py 2>/dev/null > gpurun_out/r05_v/gemm_fp8_bench.txt
python bench.py --fp8 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
p = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(p['value'], p['ms_per_step'], {n: (v['ms_per_step'], v['tflops']) for n, v in p['kernels'].items() if n.startswith('gemm256f8')})" > gpurun_out/r05_v/bench_fp8.txt
cat gpurun_out/r05_v/pytest.txt; tail -14 gpurun_out/r05_v/gemm_fp8_bench.txt; cat gpurun_out/r05_v/bench_fp8.txt
