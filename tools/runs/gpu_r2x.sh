R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2x; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 600 python tools/var_check.py 0 4 > $O/check.log 2>&1; tail -1 $O/check.log
timeout 600 python tools/gemm_stamps.py 2>&1 | grep "fc1" | sed 's/; launch span.*//'
MJV_BENCH_ROUNDS=5 timeout 900 python tools/gemm_bench.py 0 > $O/gb.log 2>&1; grep "fc1 \|qkv \|square8k" $O/gb.log
