R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ag; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=" $O/pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.log
timeout 900 python3 bench.py --pairs 8 --no-cpu-baseline --no-latency > $O/bench_c3.json 2> $O/bench_c3.log
timeout 900 python3 bench.py --image-size 224 --no-cpu-baseline --no-latency > $O/bench_c1.json 2> $O/bench_c1.log
for f in default c3 c1; do python3 -c "
import json
d=json.loads(open('gpurun_out/r3ag/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline'].get('traffic'), d.get('latency', {}).get('one_video_per_forward_ms'))"; done
