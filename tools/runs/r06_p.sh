#!/bin/bash
# round 6, run P: bench.py after its last edits (IPC default, 4B traffic file): default line, 4B line, and the launcher path at --gpus 1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_p
python bench.py > gpurun_out/r06_p/bench_default.json 2> gpurun_out/r06_p/default.err
python bench.py --backbone 4b > gpurun_out/r06_p/bench_4b.json 2> gpurun_out/r06_p/4b.err
MJV_BENCH_FORCE_LAUNCHER=1 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06_p/bench_launcher.json 2> gpurun_out/r06_p/launcher.err
python -m pytest tests -m gpu -q -k "bench" 2>&1 | tail -3
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_p/bench_*.json')):
    try:
        p = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'NO JSON', e); continue
    r = p.get('roofline', {})
    print(f, p.get('value'), p.get('ms_per_step'), p.get('n_gpus'), p.get('ranks_seen'), r.get('kernel'), r.get('frac'), r.get('traffic'), r.get('algorithmic_bytes_per_launch'), (p.get('cpu_baseline') or {}).get('value'))
PY
tail -c 400 gpurun_out/r06_p/launcher.err
