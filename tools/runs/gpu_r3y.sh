R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
MJV_BENCH_TAILS=1 MJV_BENCH_ROUNDS=3 timeout 900 python tools/gemm_bench.py 1000 1010 2>&1 | grep -E "tail_" 
