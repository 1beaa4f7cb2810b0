R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r3x; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
bash tools/collect_profiles.sh r03e > gpurun_out/r3x_collect.log 2>&1
python3 -c "
import json
d=json.loads(open('gpurun_out/prof_r03e/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
