R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ad; mkdir -p $O
cd $R
MJV_BENCH_WSTD=0.02 MJV_BENCH_ROUNDS=3 timeout 900 python tools/gemm_bench.py 2000 2002 2003 2004 2005 2006 2008 2010 2012 2016 2>/dev/null | tee $O/gm_sweep.txt
