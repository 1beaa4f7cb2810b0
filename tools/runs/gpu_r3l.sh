R=$GRAFT_REPO_ROOT; cd $R
bash tools/collect_profiles.sh r03a > gpurun_out/r3l_collect.log 2>&1
O=$R/gpurun_out/prof_r03a
cd $R
timeout 600 python3 bench.py --image-size 224 --no-cpu-baseline --no-latency > $O/bench_c1_224.json 2> $O/bench_c1.log
timeout 600 python3 bench.py --pairs 8 --no-cpu-baseline --no-latency > $O/bench_c3_shard.json 2> $O/bench_c3.log
timeout 900 python3 bench.py --frames 112 --pairs 1 --no-cpu-baseline --no-latency > $O/bench_c4.json 2> $O/bench_c4.log
timeout 600 python3 tools/attn_stamps.py > $O/attn_stamps.txt 2>&1
timeout 600 python3 tools/attn_bench.py 50 3 0,5 x > $O/attn_bench.txt 2>&1
tail -3 gpurun_out/r3l_collect.log; cat $O/pmc_summary.txt | head -20; tail -30 $O/attn_stamps.txt
for f in c1_224 c3_shard c4; do python3 -c "
import json,sys
d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
