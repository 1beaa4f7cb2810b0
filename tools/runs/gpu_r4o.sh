R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4o; mkdir -p $O
cd $R
timeout 300 python bench.py --image-size 224 --steps 20 --warmup 3 --no-cpu-baseline --no-latency > $O/bench_c1_224.json 2>$O/err.txt
timeout 300 python bench.py --pairs 8 --steps 10 --warmup 2 --no-cpu-baseline --no-latency > $O/bench_c3_shard_16_videos.json 2>>$O/err.txt
timeout 600 python bench.py --frames 112 --pairs 1 --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $O/bench_c4_112_tiles.json 2>>$O/err.txt
timeout 600 python bench.py --fp8 --frames 112 --pairs 1 --steps 5 --warmup 1 > $O/bench_c4_112_tiles_fp8.json 2>>$O/err.txt
for f in bench_c1_224 bench_c3_shard_16_videos bench_c4_112_tiles bench_c4_112_tiles_fp8; do python -c "
import json;d=json.load(open('$O/$f.json'));k=d['kernels'];print('$f',d['value'],'pairs/s',d['ms_per_step'],'ms', {n:(v['ms_per_step'],v['tflops']) for n,v in list(k.items())[:4]})"; done
