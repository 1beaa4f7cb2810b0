R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; mkdir -p $O
cd $R
timeout 600 python tools/gemm_bench.py 0 1005 > $O/gemm_skew.txt 2>&1; echo "rc=$?"
cat $O/gemm_skew.txt
