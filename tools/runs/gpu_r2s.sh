R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2s; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1
timeout 600 python tools/gemm_stress.py > $O/stress.log 2>&1; echo "stress rc=$?"; tail -1 $O/stress.log
python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2s/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d.get('latency'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:10]: print(f"{k:26s} {v['ms_per_step']:7.3f} {v.get('tflops')}")
PY
