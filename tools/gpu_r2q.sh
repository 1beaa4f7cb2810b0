R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2q; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k gemm > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for i in 1 2; do python bench.py --no-latency > $O/bench$i.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r2q/bench$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:8]: print(f"{k:26s} {v['ms_per_step']:7.3f} {v.get('tflops')}")
PY
done
