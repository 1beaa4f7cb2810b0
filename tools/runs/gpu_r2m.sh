R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2m; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1
timeout 600 python tools/gemm_stamps.py 2>&1 | grep "vit_proj\|vit_fc2\|llm_w2" | cut -c1-200
timeout 600 python tools/gemm_bench.py 0 2>&1 | grep -v amdgpu
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency > $O/bench.json 2>$O/bench.log
python - $O/bench.json <<'PY'
import json,sys
b=json.load(open(sys.argv[1]))
print(b['value'], b['ms_per_step']); print({n:(v['ms_per_step'],v['tflops']) for n,v in b['kernels'].items() if v['ms_per_step']>5})
PY
