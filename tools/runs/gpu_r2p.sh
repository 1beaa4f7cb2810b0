R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1
python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], d.get('latency'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:18]: print(f"{k:26s} {v['ms_per_step']:7.3f} {v.get('tflops')}")
PY
