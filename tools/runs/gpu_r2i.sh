R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > $O/pytest_k.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_k.log | tail -1
timeout 300 python tools/attn_bench.py 20 3 0 > $O/attn.txt 2>&1; grep variant $O/attn.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency > $O/bench.json 2>$O/bench.log
python - $O/bench.json <<'PY'
import json,sys
b=json.load(open(sys.argv[1]))
print(b['value'], b['ms_per_step']); print({n:v['ms_per_step'] for n,v in b['kernels'].items() if v['ms_per_step']>0.5})
PY
