R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2u; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 1200 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q > $O/pytest2.log 2>&1; echo "e2e rc=$?"; tail -2 $O/pytest2.log
timeout 600 python tools/attn_bench.py 20 3 0,4 x > $O/ab.log 2>&1; grep variant $O/ab.log | tail -8
python bench.py --no-latency > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2u/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:8]: print(f"{k:26s} {v['ms_per_step']:7.3f} {v.get('tflops')}")
PY
