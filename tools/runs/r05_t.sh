#!/bin/bash
# round 5, run T: 200 back-to-back steps (not a warm-clock artefact), the C1 shape (8 frames @224^2), 8 pairs per GPU through the forced launcher path
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_t
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-latency --no-secondary > gpurun_out/r05_t/bench_200_steps.json 2> gpurun_out/r05_t/err.txt
python bench.py --image-size 224 --steps 30 --warmup 5 --no-cpu-baseline --no-latency --no-secondary > gpurun_out/r05_t/bench_c1_224.json 2>> gpurun_out/r05_t/err.txt
MJV_BENCH_FORCE_LAUNCHER=1 python bench.py --gpus 1 --pairs 8 --steps 10 --warmup 2 --no-cpu-baseline --no-latency > gpurun_out/r05_t/bench_launcher_path_8_pairs.json 2>> gpurun_out/r05_t/err.txt
for f in gpurun_out/r05_t/*.json; do python - "$f" <<'PY'
import json, sys
p = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], p['value'], p['ms_per_step'], p['process_group'], p['ranks_seen'], p['ms_per_step_by_rank'], p['config']['prefix_cache_hits'])
PY
done
