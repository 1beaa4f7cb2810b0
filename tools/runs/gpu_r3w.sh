R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -k "attention or c4 or long" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python3 bench.py --frames 112 --pairs 1 --no-cpu-baseline --no-latency > $O/bench_c4.json 2> $O/bench_c4.log
python3 -c "
import json
d=json.loads(open('gpurun_out/r3w/bench_c4.json').read().strip().splitlines()[-1]); print('c4', d['value'], d['ms_per_step']); print({k:v for k,v in d['kernels'].items() if k.startswith('attn')})"
