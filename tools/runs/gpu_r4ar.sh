R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ar; mkdir -p $O
cd $R
MJV_BENCH_TAILS=1 MJV_BENCH_WSTD=0.02 MJV_BENCH_ROUNDS=4 timeout 600 python tools/gemm_bench.py 5000 5128 5256 2>/dev/null | grep "b1_" | tee $O/b1_tiles.txt
