R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gemm.log
tail -4 $O/pytest_gemm.log
MJV_BENCH_ROUNDS=3 timeout 900 python tools/gemm_bench.py 1000 1009 > $O/gemm_bench.txt 2>&1
cat $O/gemm_bench.txt
