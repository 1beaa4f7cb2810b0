R=$GRAFT_REPO_ROOT; cd $R
bash tools/collect_profiles.sh r03c > gpurun_out/r3s_collect.log 2>&1
tail -3 gpurun_out/r3s_collect.log
python3 -c "
import json
d=json.loads(open('gpurun_out/prof_r03c/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'])"
