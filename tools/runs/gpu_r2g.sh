R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "rope or gemm_bias" > $O/pytest_rope.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_rope.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "tiny or full_c2 or deterministic or streams" > $O/pytest_e2e.log 2>&1; echo "pytest e2e rc=$?"
tail -3 $O/pytest_e2e.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2>$O/bench.log
python - $O/bench.json <<'PY'
import json,sys
b=json.load(open(sys.argv[1]))
print(b['value'], b['ms_per_step']); print({n:v['ms_per_step'] for n,v in b['kernels'].items()})
PY
