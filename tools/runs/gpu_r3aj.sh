R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3aj; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=" $O/pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.log; python3 -c "
import json
d=json.loads(open('gpurun_out/r3aj/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['latency'])"
