R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O
cd $R
timeout 600 python bench.py --image-size 224 --steps 20 --warmup 3 --no-cpu-baseline --no-latency > $O/bench_c1.json 2>$O/c1.log; echo "c1 rc=$?"
timeout 600 python bench.py --pairs 8 --steps 6 --warmup 2 --no-cpu-baseline --no-latency > $O/bench_c3_shard.json 2>$O/c3.log; echo "c3 rc=$?"
timeout 900 python bench.py --frames 112 --pairs 1 --steps 4 --warmup 1 --no-cpu-baseline --no-latency > $O/bench_c4.json 2>$O/c4.log; echo "c4 rc=$?"
for f in c1 c3_shard c4; do python - $O/bench_$f.json <<'PY'
import json,sys
b=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], b['value'], b['ms_per_step'], b['config']['workload'], {k:v['ms_per_step'] for k,v in list(b['kernels'].items())[:6]})
PY
done
