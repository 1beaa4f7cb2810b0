R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2y; mkdir -p $O
cd $R
python bench.py --image-size 224 --no-cpu-baseline --no-latency > $O/bench_c1_224.json 2> $O/e1.log
python bench.py --pairs 8 --no-cpu-baseline --no-latency > $O/bench_c3_shard_16_videos.json 2> $O/e2.log
python bench.py --pairs 1 --frames 112 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/bench_c4_112_tiles.json 2> $O/e3.log
for f in bench_c1_224 bench_c3_shard_16_videos bench_c4_112_tiles; do python - <<PY
import json
d=json.loads(open('gpurun_out/r2y/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d['config'])
PY
done
