R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"
tail -4 $O/pytest_gemm.log
for code in 0 7200 0 7200; do
  if [ $code = 0 ]; then timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$code.json 2>$O/bench.log
  else timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --gemm-code $code > $O/bench_$code.json 2>$O/bench.log; fi
  python - $O/bench_$code.json <<'PY'
import json,sys
b=json.load(open(sys.argv[1]))
k=b['kernels']
print(sys.argv[1].split('_')[-1], b['value'], b['ms_per_step'], {n:v['ms_per_step'] for n,v in k.items() if n.startswith('gemm128') or n.startswith('gemm64')})
PY
done
