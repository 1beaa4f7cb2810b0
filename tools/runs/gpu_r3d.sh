# full GPU test suite + default bench line on the current build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3d/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'])
for k,v in sorted(d['kernels'].items(), key=lambda x:-x[1]['ms_per_step'])[:12]:
    print(f"{k:22s} {v['ms_per_step']:8.3f} {v.get('tflops')}")
PY
