#!/bin/bash
# round 6, run M: full kernel tables of the other BASELINE configs on the final tree (C1 @224, C3 shard of 8 pairs, C4 112 tiles)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_m
python bench.py --frames 112 --pairs 1 --no-cpu-baseline > gpurun_out/r06_m/bench_c4_112_tiles.json 2> gpurun_out/r06_m/c4.err
python bench.py --pairs 8 --no-cpu-baseline > gpurun_out/r06_m/bench_c3_shard_8_pairs.json 2> gpurun_out/r06_m/c3.err
python bench.py --image-size 224 --no-cpu-baseline > gpurun_out/r06_m/bench_c1_224.json 2> gpurun_out/r06_m/c1.err
tail -c 300 gpurun_out/r06_m/*.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_m/bench_*.json')):
    try:
        p = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'NO JSON', e); continue
    print(f, {k: p.get(k) for k in ("value", "ms_per_step", "frac_of_mfma_roofline")}, p.get('roofline', {}).get('kernel'), p.get('roofline', {}).get('frac'))
    for k, v in list(p.get('kernels', {}).items())[:14]:
        print('   ', k, v)
PY
