R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ag; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo rc=$?
python - <<'PY'
import json,os
d=json.load(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r4ag/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline'])
print(d['cpu_baseline'])
PY
