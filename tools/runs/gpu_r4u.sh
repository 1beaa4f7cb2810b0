R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O
cd $R
for rep in 1 2; do
for L in bench bench_x1 bench_x2 bench_x3; do
MJV_ATTN_MODE2=1 MJV_LIBRARY=$R/mj-video_amd/libmjv_hip_$L.so timeout 300 python tools/attn_bench.py 20 3 0 2>/dev/null | grep "m2" | grep -v 28810 | sort | awk -v L=$L '{print L, $0}' | sort -k4,4 -k5,5n | awk '!seen[$4$5]++' | tee -a $O/attn_ab.txt
done
done
